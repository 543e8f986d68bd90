"""The GPU stages of the video -> 4D pipeline chained end to end (context encoder -> Stage I AR windows -> Stage II windows)
on tiny random-init models: shapes, finiteness, every frame decoded, anchor mesh kept.  (Numerical parity of each stage
against its oracle lives in test_image_encoder.py, test_denoiser_gpu.py and test_autoencoder.py.)"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
def test_pipeline_stages_chain_on_tiny_models():
    import e2e_synthetic as E
    dev = torch.device("cuda:0")
    out = E.run(frames=6, steps=2, vertices=300, tiny=True, dev=dev)      # 2 AR windows of 4 frames (slide 3)
    assert out["unit"] == "s" and out["value"] > 0
    assert set(out["seconds"]) == {"context_encoder", "stage_I", "stage_II", "model_build_and_upload"}
    assert "2 AR window(s) of 4" in out["config"]["workload"]
