"""amcontrast3d_amd.augment.S3DISTrainAugment (csrc/augment.hip: the loader's training transforms for a whole batch in two
launches) against (a) what the reference's own transform classes produced for the fixture cloud with the fixture's draws
(tests/golden/augment_s3dis.npz) and (b) the oracle on a full-size batch with fresh draws."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _draws_of(g, tag):
    return {"contrast": torch.tensor([float(g[f"{tag}/contrast_u"]) < float(g[f"{tag}/p_contrast"])]),
            "blend": torch.tensor([float(g[f"{tag}/blend"]) if f"{tag}/blend" in g else 0.0]),
            "scale_u": torch.from_numpy(g[f"{tag}/scale_u"]).reshape(1, 3), "theta": torch.from_numpy(g[f"{tag}/theta"]).reshape(1, 3),
            "noise": torch.from_numpy(g[f"{tag}/noise"]).unsqueeze(0),
            "drop": torch.tensor([float(g[f"{tag}/drop_u"].reshape(-1)[0]) < float(g[f"{tag}/p_drop"])])}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_matches_the_reference_run(tag):
    from amcontrast3d_amd.augment import S3DISTrainAugment
    g = load_golden("augment_s3dis")
    aug = S3DISTrainAugment(**g["meta"])
    d = {k: v.to(DEV) for k, v in _draws_of(g, tag).items()}
    pos, x, h = aug(torch.from_numpy(g["coord"]).unsqueeze(0).to(DEV), torch.from_numpy(g["feat"]).unsqueeze(0).to(DEV), draws=d)
    np.testing.assert_allclose(pos[0].cpu().numpy(), g[f"{tag}/pos"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(x[0].cpu().numpy(), g[f"{tag}/x"], rtol=0, atol=2e-5)
    np.testing.assert_array_equal(h[0].cpu().numpy(), g[f"{tag}/heights"])


def test_full_size_batch_against_the_oracle():
    from amcontrast3d_amd import synthetic
    from amcontrast3d_amd.augment import S3DISTrainAugment
    from oracle import augment_ref
    nb = synthetic.make_batch(8, 24000, first_id=60)
    pos = torch.from_numpy(nb["pos"]).to(DEV)
    color = torch.from_numpy(np.ascontiguousarray(nb["x"][:, :3].transpose(0, 2, 1)) * 255.0).float().to(DEV)
    aug = S3DISTrainAugment(contrast_p=0.5, color_drop=0.3)
    gen = torch.Generator(device=DEV).manual_seed(3)
    d = aug.draw(8, 24000, torch.device(DEV), gen)
    assert bool(d["contrast"].any()) and bool((~d["contrast"]).any())
    po, xo, ho = aug(pos, color, draws=d)
    again = aug(pos, color, draws=d)
    assert all(torch.equal(a, b) for a, b in zip((po, xo, ho), again)), "deterministic"
    for b in range(8):
        dd = {"contrast": bool(d["contrast"][b]), "blend": float(d["blend"][b]), "scale_u": d["scale_u"][b].cpu().numpy(),
              "theta": d["theta"][b].cpu().numpy(), "noise": d["noise"][b].cpu().numpy(), "drop": bool(d["drop"][b])}
        wp, wx, wh = augment_ref.s3dis_train(nb["pos"][b], color[b].cpu().numpy(), dd)
        np.testing.assert_allclose(po[b].cpu().numpy(), wp, rtol=0, atol=3e-6)
        np.testing.assert_allclose(xo[b].cpu().numpy(), wx, rtol=0, atol=3e-5)
        np.testing.assert_array_equal(ho[b].cpu().numpy(), wh)
    # what the chain guarantees: centred in xy, resting on z = 0 (before the jitter of at most `clip`), colours normalised
    assert float(po[..., 2].min()) >= -0.02 - 1e-6 and abs(float(po[..., :2].mean())) < 0.05
    # fresh draws from a generator: runs, and differs from cloud to cloud
    p2, _, _ = aug(pos, color, generator=gen)
    assert not torch.equal(p2[0], p2[1])
