"""dataset.create_spec (native path) + split: oracle vs reference-generated goldens (CPU) and the HIP
re-framing kernel vs the oracle (GPU). SURVEY.md section 8f, row f1; Postprocessor kernel (row f2)."""

import importlib

import numpy as np
import pytest
import torch

from conftest import PKG
from oracle import dataset_frames as ods

CLIPS = (("clip4s", 64000), ("clip2p5s", 40000), ("clip1s", 16000))


def test_oracle_matches_reference_dataset_functions(golden, W):
    g = golden("dataset")
    for name, n in CLIPS:
        wav = W.uniform(31, W.stream_id("dataset/" + name), n, dtype=np.float64)
        spec = ods.create_spec_native(wav)
        assert np.array_equal(spec.astype(np.float32), g["spec/" + name])
        assert np.array_equal(ods.split(spec, 10, 96, 64, True).astype(np.float32), g["frames_overlap/" + name])
        assert np.array_equal(ods.split(spec, 10, 96, 64, False).astype(np.float32), g["frames_contig/" + name])
    with pytest.raises(ValueError):
        ods.create_spec_native(np.zeros(80000))          # 5 examples do not fit 4 slots (dataset.py:321-322)


@pytest.mark.gpu
def test_hip_reframing_matches_reference(golden, W):
    g = golden("dataset")
    ds = importlib.import_module(PKG + ".dataset")
    for name, n in CLIPS:
        wav = W.uniform(31, W.stream_id("dataset/" + name), n, dtype=np.float64)
        spec = ds.create_spec(wav, "vggish", 16000, 64000, 96, 64, False, True)
        ref = g["spec/" + name]
        assert tuple(spec.shape) == (64, 384)
        pad = ref == 0.0
        assert np.array_equal(spec.cpu().numpy() == 0.0, pad), "zero-padded slots are exact"
        assert np.abs(spec.cpu().numpy() - ref).max() <= 1e-4
        for overlap, key in ((True, "frames_overlap/"), (False, "frames_contig/")):
            fr = ds.split(spec, 10, 96, 64, overlap)
            assert np.array_equal(fr.cpu().numpy(), ds.split(torch.from_numpy(spec.cpu().numpy()), 10, 96, 64, overlap).numpy())
            assert fr.shape == g[key + name].shape and np.abs(fr.cpu().numpy() - g[key + name]).max() <= 1e-4
    # batched path: bit-identical to per-clip create_spec + split
    waves = np.stack([W.uniform(31, W.stream_id("dataset/clip4s"), 64000), W.uniform(32, 5, 64000)])
    batch = ds.clips_to_frames(torch.from_numpy(waves).cuda())
    assert tuple(batch.shape) == (2, 10, 1, 64, 96)
    for c in range(2):
        spec = ds.create_spec(waves[c].astype(np.float64), "vggish", 16000, 64000, 96, 64, False, True)
        assert torch.equal(batch[c, :, 0], ds.split(spec, 10, 96, 64, True))
    with pytest.raises(ValueError):
        ds.create_spec(np.zeros(80000), "vggish", 16000, 64000, 96, 64, False, True)
    with pytest.raises(NotImplementedError):
        ds.create_spec(np.zeros(64000), "vggish", 16000, 64000, 96, 64, True, True)
