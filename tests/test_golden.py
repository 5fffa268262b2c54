"""Golden-vector tests (tests/golden/*.npz, written by tests/golden/make_golden.py from the oracle).

CPU half: the oracle still reproduces its own committed vectors bit for bit (guards the restatement against
accidental edits).  GPU half: the HIP path against the same vectors WITHOUT importing the oracle."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = lambda name: np.load(os.path.join(HERE, "golden", name))

TOL_H_REL, TOL_UPD_REL, TOL_UPD_ABS = 3e-5, 2e-4, 2e-7


# ------------------------------------------------------------------ oracle vs its committed vectors (CPU)
def test_oracle_reproduces_gn_step_goldens():
    import orc
    z = G("gn_step.npz")
    for level in (1, 2):
        for j in (0, 1):
            k = "L%d_p%d_" % (level, j)
            o = orc.optimize(z["L%d_obj" % level], z["L%d_ref" % level], z["L%d_depth" % level], z["L%d_sigma" % level],
                             z["L%d_K" % level], z[k + "xi"], level, want_mask=True)
            np.testing.assert_array_equal(np.packbits(o["mask"]), z[k + "mask"])
            np.testing.assert_array_equal(o["H"], z[k + "H"])
            np.testing.assert_array_equal(o["g"], z[k + "g"])
            np.testing.assert_array_equal(o["xi_update"], z[k + "xi_update"])
            assert o["n_valid"] == int(z[k + "n_valid"])


def test_oracle_reproduces_track_image_and_mapping_goldens():
    import orc
    z = G("track.npz")
    r = orc.OFrame(z["ref"], z["depth"], z["sigma"], z["K"], 3, 0)
    o = orc.OFrame(z["obj"], None, None, z["K"], 3, 0)
    xi, log = orc.track(o, r)
    np.testing.assert_array_equal(xi, z["xi"])
    assert log["n_iter"] == z["n_iter"].tolist()
    z = G("image_ops.npz")
    np.testing.assert_array_equal(orc.warp_image(z["xi"], z["gray"], z["depth"], z["K"]), z["warped"])
    np.testing.assert_array_equal(orc.gradiate(z["gray"], True), z["gradx"])
    np.testing.assert_array_equal(orc.cull_image(z["gray"], 2), z["cull2"])
    z = G("mapping.npz")
    pd, ps, pa = orc.propagate(z["depth"], z["sigma"], z["age"], z["rel"], z["K"])
    np.testing.assert_array_equal(pd, z["prop_depth"]); np.testing.assert_array_equal(pa, z["prop_age"])
    np.testing.assert_array_equal(orc.regularize(z["depth"], z["sigma"]), z["regularized"])


# ------------------------------------------------------------------ HIP path vs the committed vectors (GPU)
@pytest.mark.gpu
def test_gpu_gn_step_matches_goldens():
    import dvo_amd as dvo
    z = G("gn_step.npz")
    for level in (1, 2):
        for j in (0, 1):
            k = "L%d_p%d_" % (level, j)
            r = dvo.optimize(z["L%d_obj" % level], z["L%d_ref" % level], z["L%d_depth" % level], z["L%d_sigma" % level],
                             z["L%d_K" % level], z[k + "xi"], level, want_mask=True)
            np.testing.assert_array_equal(np.packbits(r["mask"]), z[k + "mask"])       # pixel selection: bit exact
            assert r["n_valid"] == int(z[k + "n_valid"])
            np.testing.assert_allclose(r["H"], z[k + "H"], rtol=0, atol=TOL_H_REL * np.abs(z[k + "H"]).max())
            np.testing.assert_allclose(r["g"], z[k + "g"], rtol=0, atol=TOL_H_REL * np.abs(z[k + "g"]).max())
            un = np.linalg.norm(z[k + "xi_update"])
            np.testing.assert_allclose(r["xi_update"], z[k + "xi_update"], rtol=0, atol=TOL_UPD_REL * un + TOL_UPD_ABS)
            np.testing.assert_allclose(r["residual"], z[k + "residual"], rtol=2e-5)


@pytest.mark.gpu
def test_gpu_track_matches_golden():
    import dvo_amd as dvo
    z = G("track.npz")
    xi, log = dvo.track(z["obj"], z["ref"], z["depth"], z["sigma"], z["K"], 3, 0)
    assert log["n_iter"] == z["n_iter"].tolist()
    np.testing.assert_array_equal(np.concatenate(log["n_valid"]), z["n_valid"])
    np.testing.assert_allclose(np.concatenate(log["residual"]), z["residual"], rtol=1e-4)
    np.testing.assert_allclose(np.concatenate(log["xi_after"]), z["xi_after"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(xi, z["xi"], rtol=0, atol=2e-5)


@pytest.mark.gpu
def test_gpu_image_ops_match_goldens_bit_exact():
    import dvo_amd as dvo
    z = G("image_ops.npz")
    np.testing.assert_array_equal(dvo.Transform.warpImage(z["xi"], z["gray"], z["depth"], z["K"]), z["warped"])
    np.testing.assert_array_equal(dvo.Convert.gradiate(z["gray"], True), z["gradx"])
    np.testing.assert_array_equal(dvo.Convert.gradiate(z["gray"], False), z["grady"])
    np.testing.assert_array_equal(dvo.Convert.cullImage(z["gray"], 1), z["cull1"])
    np.testing.assert_array_equal(dvo.Convert.cullImage(z["gray"], 2), z["cull2"])


@pytest.mark.gpu
def test_gpu_mapping_matches_goldens_bit_exact():
    import dvo_amd as dvo
    z = G("mapping.npz")
    pd, ps, pa = dvo.Implement.propagate(z["depth"], z["sigma"], z["age"], z["rel"], z["K"])
    np.testing.assert_array_equal(pd, z["prop_depth"])
    np.testing.assert_array_equal(ps, z["prop_sigma"])
    np.testing.assert_array_equal(pa, z["prop_age"])
    np.testing.assert_array_equal(dvo.Implement.regularize(z["depth"], z["sigma"]), z["regularized"])
    d, s, a, v = dvo.mapper_update(list(z["hist_gray"]), z["hist_xi"], z["obj_gray"], z["obj_xi"], z["obj_rel"], int(z["obj_id"]),
                                   z["K"], z["depth"], z["sigma"], z["age"], cfg=dvo.default_config(rng_seed=int(z["seed"])))
    np.testing.assert_array_equal(d, z["upd_depth"])
    np.testing.assert_array_equal(s, z["upd_sigma"])
    np.testing.assert_array_equal(a, z["upd_age"])
    assert v == int(z["valid"])
