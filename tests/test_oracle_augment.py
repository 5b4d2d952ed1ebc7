"""oracle/augment_ref.py (numpy restatement of the S3DIS training transforms) against what the reference's own transform
classes produced for the same cloud and the same random draws (tests/golden/augment_s3dis.npz, oracle/gen_golden.py augment)."""
import numpy as np
import pytest

from conftest import load_golden


@pytest.mark.parametrize("tag", ["a", "b"])
def test_augment_oracle_matches_reference_run(tag):
    from oracle import augment_ref
    g = load_golden("augment_s3dis")
    d = {"contrast": float(g[f"{tag}/contrast_u"]) < float(g[f"{tag}/p_contrast"]),
         "blend": float(g[f"{tag}/blend"]) if f"{tag}/blend" in g else 0.0,
         "scale_u": g[f"{tag}/scale_u"], "theta": g[f"{tag}/theta"], "noise": g[f"{tag}/noise"],
         "drop": float(g[f"{tag}/drop_u"].reshape(-1)[0]) < float(g[f"{tag}/p_drop"])}
    assert d["contrast"] == (tag == "a") and d["drop"] == (tag == "a")
    pos, x, heights = augment_ref.s3dis_train(g["coord"], g["feat"], d)
    np.testing.assert_allclose(pos, g[f"{tag}/pos"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(x, g[f"{tag}/x"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(heights, g[f"{tag}/heights"], rtol=0, atol=1e-6)
    assert np.array_equal(heights[:, 0], g["coord"][:, 2]) and abs(float(pos[:, :2].mean())) < 2e-2
