#!/bin/bash
# A/B of the u8-tap experiment on one box (bash tools/experiments/ab_u8taps.sh): the same experimental build with the u8 taps on and off
# (DVO_EXP_U8TAPS_OFF=1): bit-identity of a raw-frame batch, finest-level probe, full bench, PMC of the probe.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; export DVO_LIB_PATH=$R/ab/libdvo_u8taps.so
python3 - <<'PY'
import os, sys
sys.path.insert(0, "direct-visual-odometry_amd")
import numpy as np, torch
import dvo_amd as dvo
from dvo_amd import synth
dev = torch.device("cuda", 0)
K = synth.K_640
fr = [synth.render(p, K, 640, 480, device=dev) for p in synth.trajectory(3, seed=42)]
g8 = [(f[0].clamp(0, 1) * 255).round().to(torch.uint8).unsqueeze(0).contiguous() for f in fr]
d16 = [(f[1].clamp(0, 13) * 5000).round().to(torch.int32).to(torch.int16).unsqueeze(0).contiguous() for f in fr]
out = {}
for off in ("", "1"):
    if off: os.environ["DVO_EXP_U8TAPS_OFF"] = "1"
    else: os.environ.pop("DVO_EXP_U8TAPS_OFF", None)
    bt = dvo.Batch(1, K, 640, 480, 4, 1, cfg=dvo.default_config(gn_pixels_per_thread=4))
    xs = []
    for k in range(3):
        bt.push_raw_device(g8[k].data_ptr(), 1, d16[k].data_ptr())
        if k: xs.append(bt.last_poses()[0].copy())
    out[off] = np.stack(xs); bt.close()
print("u8 taps vs float taps, poses bit-identical:", np.array_equal(out[""].view(np.uint32), out["1"].view(np.uint32)), out[""][0])
PY
for rep in 1 2; do
  echo -n "probe u8 taps ON : "; python3 tools/probe_gn.py --raw --batch 256 --level 3 --launches 20 --sigma 0.1 --ppt 4 --group 2 2>/dev/null | tail -1
  echo -n "probe u8 taps OFF: "; DVO_EXP_U8TAPS_OFF=1 python3 tools/probe_gn.py --raw --batch 256 --level 3 --launches 20 --sigma 0.1 --ppt 4 --group 2 2>/dev/null | tail -1
done
for v in ON OFF; do
  if [ $v = OFF ]; then export DVO_EXP_U8TAPS_OFF=1; else unset DVO_EXP_U8TAPS_OFF; fi
  python3 bench.py --no-cpu-baseline --pcie-steps 0 --no-secondary --batch 4096 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench u8 taps $v:', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms/step, gn avg', round(d['roofline']['avg_launch_us'],1), 'us')"
done
unset DVO_EXP_U8TAPS_OFF
PROBE_EXTRA=--raw bash tools/pmc_gn.sh 0.1 4 2 256 > /dev/null 2>&1; cp /tmp/pmc_all.txt gpurun_out/r03_exp_u8taps_pmc_raw_b256.txt
grep -A24 "k_track_gn<4, 2, false, true>" gpurun_out/r03_exp_u8taps_pmc_raw_b256.txt | egrep "avg_ns|SQ_INSTS_VALU |SQ_INSTS_VMEM_RD|TD_TD_BUSY|TA_TA_BUSY|TCP_PENDING|SQ_ACTIVE_INST_VALU|GRBM_GUI|TCP_TOTAL_CACHE" 
