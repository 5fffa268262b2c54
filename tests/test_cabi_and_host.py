"""CPU tests of the drop-in boundary: libdvo.so loads, exports every symbol include/dvo.h declares, and FAILS
LOUDLY (no CPU fallback) when asked to compute without a GPU.  No compute call succeeds here by design."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import dvo_amd as dvo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "dvo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dvo_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = dvo.lib()
    names = _header_functions()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(dvo.EXPORTS) == names          # the Python binding list and the header agree


def test_library_exports_nothing_but_the_c_abi():
    # The engine's C++ types live in namespace dvo, like the facade's (include/dvo.hpp): an exported dvo::Keyframe::~Keyframe() of
    # the library is resolved to the host PROGRAM's dvo::Keyframe by the dynamic linker (seen: a segfault in free()).
    so = os.path.join(ROOT, "direct-visual-odometry_amd", "lib", "libdvo.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert names and all(n.startswith("dvo_") for n in names), [n for n in names if not n.startswith("dvo_")][:10]
    assert sorted(names) == _header_functions()


def test_config_defaults_are_the_reference_literals():
    c = dvo.default_config()
    assert c.max_iterations == 15 and abs(c.min_update - 5e-4) < 1e-9 and abs(c.min_residual - 5e-3) < 1e-9  # tracker.cpp:16-19
    assert (c.step_default, c.step_level1, c.step_level2) == (2.0, 1.5, 1.0)                                # optimize.cpp:22-26
    assert abs(c.sigma_min - 0.01) < 1e-9 and c.sigma_max == 0.5 and abs(c.min_depth - 0.2) < 1e-7          # optimize.cpp:39,83
    assert abs(c.keyframe_min_translation - 0.02) < 1e-9 and c.keyframe_max_frames == 6                     # mapper.cpp:12-13
    assert c.crop_enable == 1 and c.fixed_iterations == 0
    assert C.sizeof(dvo.Config) == 96 or C.sizeof(dvo.Config) % 8 == 0


def test_status_strings():
    L = dvo.lib()
    assert L.dvo_status_string(0) == b"ok"
    assert b"no CPU fallback" in L.dvo_status_string(3)
    assert b"gfx950" in L.dvo_version()


@pytest.mark.skipif(dvo.device_count() > 0, reason="only meaningful on a box without a GPU")
def test_compute_entry_points_fail_loudly_without_a_gpu():
    K = np.array([525, 0, 319.5, 0, 525, 239.5, 0, 0, 1], np.float32)
    with pytest.raises(dvo.DvoError, match="no HIP device|no CPU fallback"):
        dvo.VisualOdometry(K, 640, 480)
    with pytest.raises(dvo.DvoError):
        dvo.Batch(4, K, 640, 480)
    img = np.zeros((8, 8), np.float32)
    with pytest.raises(dvo.DvoError):
        dvo.Convert.cullImage(img, 1)
    with pytest.raises(dvo.DvoError):
        dvo.se3.exp(np.zeros(6, np.float32))


def test_bad_arguments_are_status_codes():
    L = dvo.lib()
    assert L.dvo_vo_create(None, 640, 480, None, None) == 1          # DVO_ERR_BAD_ARGUMENT, never abort()
    assert L.dvo_op_cull_image(0, None, 4, 4, 1, None) == 1
    assert L.dvo_batch_destroy(None) == 0 and L.dvo_vo_destroy(None) == 0
    assert L.dvo_vo_keyframe_count(None) == 0


def test_product_never_references_the_oracle():
    # the product path must not link, load or call anything under oracle/
    pkg = os.path.join(ROOT, "direct-visual-odometry_amd")
    for sub, exts in (("csrc", (".h", ".hip", ".cpp")), ("dvo_amd", (".py",))):
        for fn in os.listdir(os.path.join(pkg, sub)):
            if fn.endswith(exts):
                txt = open(os.path.join(pkg, sub, fn)).read()
                assert "liboracle" not in txt and "dvo_oracle" not in txt and "import orc" not in txt, fn
    out = os.popen("ldd %s" % dvo.LIB_PATH).read()
    assert "oracle" not in out


def test_cpp_facade_header_compiles():
    # include/dvo.hpp (System::VisualOdometry facade) must be valid C++17 against include/dvo.h
    import subprocess, tempfile
    src = '#include "dvo.hpp"\nint main(){ dvo::Mat3 K{}; (void)K; auto c = dvo::default_config(); return c.max_iterations == 15 ? 0 : 1; }\n'
    with tempfile.NamedTemporaryFile("w", suffix=".cpp", delete=False) as f:
        f.write(src)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), f.name], capture_output=True, text=True)
    os.unlink(f.name)
    assert r.returncode == 0, r.stderr


def test_hot_kernel_fits_its_register_budget():
    """k_track_gn is built for 6 waves per SIMD (up to 84 VGPRs; 7 waves = 72 until the one-round-trip border sampler, which spilled
    24 bytes there) and its time is proportional to its instruction count and its memory traffic: a spilled register is both.  The
    default variants must compile without scratch memory."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    pkg = os.path.join(ROOT, "direct-visual-odometry_amd")
    flags = None
    for line in open(os.path.join(pkg, "Makefile")):
        if line.startswith("FLAGS"):
            flags = line.split("=", 1)[1].strip().rstrip("\\").split()
    cont = open(os.path.join(pkg, "Makefile")).read().split("FLAGS   =", 1)[1].split("\n")
    flags = (cont[0].rstrip("\\") + " " + cont[1]).split()
    flags = [f.replace("$(ARCH)", "gfx950") for f in flags if f != "-fPIC"]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(pkg, "csrc", "dvo_kernels.hip")],
                       check=True, capture_output=True, timeout=900)
        txt = open(out).read()
    checked = 0
    for variant in ("ILi4ELi2ELb0ELb0E", "ILi4ELi2ELb0ELb1E"):   # <PPT 4, G 2, no mask, raster | 2-D tiles>
        m = re.search(r"\.amdhsa_kernel _ZN3dvo10k_track_gn%s.*?\.end_amdhsa_kernel" % variant, txt, re.S)
        assert m, "kernel variant not found: " + variant
        body = m.group(0)
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        assert scratch == 0, "%s spills %d bytes of scratch per lane" % (variant, scratch)
        assert vgpr <= 84, "%s needs %d VGPRs (budget 84 = 6 waves per SIMD)" % (variant, vgpr)
        checked += 1
    assert checked == 2


def test_tile_geometry_covers_every_pixel_once():
    """gn_tiling() (csrc/dvo_kernels.h) is the single source of tile counts for host and device.  For a spread of level shapes:
    every pixel that can contribute (inside the crop window on a crop level) belongs to exactly one live tile."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = r'''
#include "dvo_kernels.h"
#include <cstdio>
#include <vector>
using namespace dvo;
int main() {
    const int shapes[][3] = {{320,240,0},{160,120,1},{160,120,0},{80,60,0},{40,30,0},{1920,1080,0},{960,540,0},{480,270,0},{240,135,0},
                             {120,67,0},{322,243,0},{37,29,0},{64,16,0},{64,48,0},{16,200,0},{150,110,1},{30,25,1}};
    for (auto& sh : shapes) for (int ppt : {1, 2, 4, 8}) {
        const int w = sh[0], h = sh[1], crop = sh[2];
        const GnTiling t = gn_tiling(w, h, ppt, crop);
        if (t.live_first < 0 || t.live_count < 0 || t.live_first + t.live_count > t.count) { printf("bad live range %d %d %d\n", w, h, ppt); return 1; }
        std::vector<int> hits((size_t)w * h, 0);
        for (int b = t.live_first; b < t.live_first + t.live_count; b++) {
            if (t.t2d) {
                const int tw = 1 << t.shift, rw = 64 >> t.shift, R = rw * 4 * ppt;
                const int ty = b / t.tiles_x, tx = b % t.tiles_x;
                for (int y = t.y_org + ty * R; y < t.y_org + ty * R + R; y++)
                    for (int x = t.x_org + tx * tw; x < t.x_org + tx * tw + tw; x++)
                        if (x < w && y < h) hits[(size_t)y * w + x]++;
            } else {
                for (long long i = (long long)b * 256 * ppt; i < (long long)(b + 1) * 256 * ppt; i++)
                    if (i < (long long)w * h) hits[(size_t)i]++;
            }
        }
        long long live = 0;
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            const bool need = !crop || (x >= 20 && x <= 140 && y >= 20 && y <= 100);
            const int c = hits[(size_t)y * w + x];
            if (c > 1 || (need && c != 1)) { printf("pixel (%d,%d) of %dx%d ppt %d crop %d covered %d times\n", x, y, w, h, ppt, crop, c); return 1; }
            live += c;
        }
        if (live != t.live_pixels) { printf("live_pixels %lld != %lld for %dx%d ppt %d crop %d\n", t.live_pixels, live, w, h, ppt, crop); return 1; }
    }
    printf("ok\n");
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as td:
        cpp = os.path.join(td, "t.cpp")
        open(cpp, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.run([hipcc, "-std=c++17", "-x", "hip", "--cuda-host-only", "-I", os.path.join(ROOT, "direct-visual-odometry_amd", "csrc"),
                        cpp, "-o", exe], check=True, capture_output=True, timeout=600)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_gauss_gain_shortcut_matches_the_division():
    """gauss_gain() (csrc/dvo_math.h) replaces the reference's m / 0.8f (gaussian.cpp:20) by a multiply and one residual step.
    tools/verify/gauss_gain_div.c walks float bit patterns and compares the resulting gains bit for bit (all 2^32 when run by hand;
    every 5th here), and dvo_math.h must still contain the constants the program checks."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not available")
    math_h = open(os.path.join(ROOT, "direct-visual-odometry_amd", "csrc", "dvo_math.h")).read()
    assert "q0 = m * 1.25f" in math_h and "fmaf(fmaf(-q0, 0.8f, m), 1.25f, q0)" in math_h and "fabsf(m) < 1e30f" in math_h
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "ggd")
        subprocess.run([gcc, "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "verify", "gauss_gain_div.c"), "-lm"],
                       check=True, capture_output=True, timeout=300)
        r = subprocess.run([exe, "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 gains differ, 0 quotients differ" in r.stdout, r.stdout + r.stderr


def test_host_code_is_clean_under_address_and_undefined_sanitizers():
    """the oracle's whole path and the host-only translation units of libdvo (PNG / dataset front-end with a
    mutation fuzz and crafted headers, trajectory evaluation) run once under -fsanitize=address,undefined.  GPU ASan does
    not exist on this pool, so the kernels are covered by the parity tests instead."""
    import subprocess
    for sub in ("oracle", "direct-visual-odometry_amd"):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, sub), "asan"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "asan driver ok" in r.stdout
