"""Zero-edit drop-in (VERDICT r03 missing #2): `actionmesh_amd.install()` + `python -m actionmesh_amd.cli` against the REFERENCE's
own, unmodified `actionmesh.pipeline` and `inference/video_to_animated_mesh.py` (CPU; build container only - skipped where
/root/reference is absent).  The container lacks the reference's heavy dependencies (hydra, omegaconf, trimesh, cv2, triposg,
skimage, torchvision, natsort), so they are supplied as inert stub modules - enough for `import actionmesh.pipeline` and for the
CLI script to run up to the point where the pipeline would download / load weights.  What is asserted:

  * the patched names resolve where the reference looks them up: `_load_temporal_denoiser` (pipeline.py:171-184, the reference's
    own function object) produces a HipDenoiser from a `config.json + model.safetensors` directory;
  * the preset mapping covers all four `--fast` / `--low_ram` combinations, and the config directory the reference's `load_config`
    receives holds the reference's YAMLs and the overlays side by side, with every overlay's `defaults:` entry present;
  * the reference CLI script itself, run through `actionmesh_amd.cli` with its own argparse flags, constructs its pipeline with the
    preset it derived (unchanged) and `load_config` serves the overlay; `--backend reference` leaves everything untouched;
  * `uninstall()` restores every name.
"""
import json
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "actionmesh")), reason="reference not present")

class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (object,), {})


def _import_with_stubs(monkeypatch, module):
    """Import `module`, supplying an inert stub for every dependency this container lacks - on demand, the way the import fails, so a
    package that merely PROBES for an optional dependency (transformers looking for torchvision) still sees it as absent."""
    import importlib
    for _ in range(64):
        try:
            return importlib.import_module(module)
        except ModuleNotFoundError as e:
            parts = e.name.split(".")
            for k in range(1, len(parts) + 1):
                m = ".".join(parts[:k])
                if m not in sys.modules:
                    sm = _Stub(m)
                    sm.__path__ = []
                    monkeypatch.setitem(sys.modules, m, sm)
    raise RuntimeError(f"could not import {module}")


@pytest.fixture()
def reference_pipeline(monkeypatch):
    for p in (os.path.join(ROOT, "oracle", "diffusers_shim"), REF):
        if p not in sys.path:
            monkeypatch.syspath_prepend(p)
    P = _import_with_stubs(monkeypatch, "actionmesh.pipeline")
    for m in ("actionmesh.io.glb_export", "actionmesh.io.mesh_io", "actionmesh.io.video_input"):      # what the CLI script imports
        _import_with_stubs(monkeypatch, m)
    from actionmesh_amd import dropin
    yield P
    dropin.uninstall()


def _write_checkpoint(tmp_path):
    """<dir>/denoiser/{config.json, model.safetensors}: the PyTorchModelHubMixin layout pipeline.py:180-182 reads."""
    from safetensors.torch import save_file
    from oracle import denoiser_oracle as O
    kw = dict(in_channels=64, num_layers=3, num_attention_heads=2, width=256, mlp_ratio=4.0, cross_attention_dim=64,
              inflated_layers=[0, 1, 2])
    sd = O.synthetic_state_dict(O.OracleConfig(**{**kw, "inflated_layers": (0, 1, 2)}), seed=1)
    d = tmp_path / "ActionMesh" / "denoiser"
    d.mkdir(parents=True)
    (d / "config.json").write_text(json.dumps(dict(kw, num_tokens_nominal=32, temporal_context_size=4)))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
    return str(tmp_path / "ActionMesh"), sd


def test_install_puts_hipdenoiser_behind_the_reference_loader(reference_pipeline, tmp_path):
    import actionmesh_amd
    from actionmesh_amd import HipDenoiser
    P = reference_pipeline
    ref_denoiser, ref_load_config = P.ActionMeshDenoiser, P.load_config
    actionmesh_amd.install(attn_dtype="fp8")
    assert issubclass(P.ActionMeshDenoiser, HipDenoiser) and P.load_config is not ref_load_config
    weights_dir, sd = _write_checkpoint(tmp_path)
    # the reference's OWN loader function, on an object that carries just the attributes it reads
    fake = types.SimpleNamespace(temporal_3D_denoiser=None, _actionmesh_weights_dir=weights_dir, device=torch.device("cpu"))
    P.ActionMeshPipeline._load_temporal_denoiser(fake)
    m = fake.temporal_3D_denoiser
    assert isinstance(m, HipDenoiser) and m.attn_dtype == "fp8" and m.num_layers == 3 and m.width == 256
    assert m.device == torch.device("cpu") and set(m._host_sd) == set(sd)
    assert all(torch.equal(m._host_sd[k], sd[k].float()) for k in sd)
    P.ActionMeshPipeline._load_temporal_denoiser(fake)           # "already loaded on this device": the early return
    assert fake.temporal_3D_denoiser is m
    actionmesh_amd.install(attn_dtype="bf16", stage2=True)       # idempotent re-install with other options
    from actionmesh_amd import HipAutoencoder
    assert P.ActionMeshAutoencoder is HipAutoencoder
    actionmesh_amd.uninstall()
    assert P.ActionMeshDenoiser is ref_denoiser and P.load_config is ref_load_config
    assert P.ActionMeshAutoencoder is not HipAutoencoder


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("low_ram", [False, True])
def test_preset_mapping_and_merged_config_dir(reference_pipeline, monkeypatch, fast, low_ram):
    import yaml
    from actionmesh_amd import dropin
    P = reference_pipeline
    seen = {}

    def fake_load_config(config_name, config_dir, updates={}):
        seen["args"] = (config_name, config_dir)
        with open(os.path.join(config_dir, config_name)) as f:
            return yaml.safe_load(f)
    monkeypatch.setattr(P, "load_config", fake_load_config)
    dropin.install()
    ref_name = dropin.preset_for_flags(fast, low_ram, backend="reference")
    ref_dir = os.path.join(REF, "actionmesh", "configs")
    assert os.path.exists(os.path.join(ref_dir, ref_name)), "the CLI's preset names (video_to_animated_mesh.py:199-210)"
    cfg = P.load_config(ref_name, ref_dir)
    name, cdir = seen["args"]
    assert name == dropin.preset_for_flags(fast, low_ram, backend="hip") == ref_name.replace(".yaml", "_mi355x.yaml")
    assert cdir != ref_dir and not os.path.samefile(cdir, ref_dir)
    have = set(os.listdir(cdir))
    assert set(fn for fn in os.listdir(ref_dir) if fn.endswith(".yaml")) <= have, "reference presets copied"
    assert {p + "_mi355x.yaml" for p in dropin.REFERENCE_PRESETS} <= have
    assert cfg["defaults"] == [ref_name[:-5]] and (ref_name in have)          # Hydra resolves the overlay's parent in the same dir
    assert cfg["model"]["scheduler"]["_target_"] == dropin.SCHEDULER_TARGET
    assert cfg["model"]["cf_guidance"]["_target_"] == dropin.GUIDANCE_TARGET
    # an overlay name passes through; a preset of the user's own gets the two targets rewritten on the composed config
    P.load_config(name, cdir)
    assert seen["args"] == (name, cdir)


def test_unknown_preset_is_retargeted(reference_pipeline, monkeypatch, tmp_path):
    from actionmesh_amd import dropin
    P = reference_pipeline
    (tmp_path / "mine.yaml").write_text("x: 1\n")
    ref_load_config = P.load_config
    monkeypatch.setattr(P, "load_config", ref_load_config)          # restored at teardown whatever the test assigns below
    P.load_config = lambda name, d, updates={}: {"model": {"scheduler": {"_target_": "actionmesh.scheduler.scheduler.SchedulerFlow", "shift": 3.0},
                                                           "cf_guidance": {"_target_": "actionmesh.scheduler.guidance.ClassifierFreeGuidance"}}}
    dropin.install()
    cfg = P.load_config("mine.yaml", str(tmp_path))
    assert cfg["model"]["scheduler"] == {"_target_": dropin.SCHEDULER_TARGET, "shift": 3.0}
    assert cfg["model"]["cf_guidance"]["_target_"] == dropin.GUIDANCE_TARGET
    dropin.uninstall()
    P.load_config = lambda name, d, updates={}: {"model": {}}
    dropin.install()
    with pytest.raises(RuntimeError, match="no model.scheduler"):
        P.load_config("mine.yaml", str(tmp_path))


class _Stop(Exception):
    pass


@pytest.mark.parametrize("flags,preset", [((), "actionmesh.yaml"), (("--fast",), "actionmesh_fast.yaml"),
                                          (("--low_ram",), "actionmesh_lowram.yaml"), (("--fast", "--low_ram"), "actionmesh_fast_lowram.yaml")])
@pytest.mark.parametrize("backend", ["hip", "reference"])
def test_reference_cli_runs_unmodified_through_the_wrapper(reference_pipeline, tmp_path, monkeypatch, flags, preset, backend):
    """The reference script, byte for byte, executed as __main__ by actionmesh_amd.cli: its own argparse, its own preset choice, its
    own `ActionMeshPipeline(config_name=..., config_dir=..., dtype=..., lazy_loading=...)` call - recorded by a stand-in pipeline
    class that then asks `load_config` for its config exactly as pipeline.py:66 does, and stops before any weight is touched."""
    from actionmesh_amd import cli, dropin
    P = reference_pipeline
    seen = {}

    def fake_load_config(config_name, config_dir, updates={}):
        seen["load_config"] = (config_name, os.path.abspath(config_dir))
        return {"model": {"scheduler": {"_target_": "x"}, "cf_guidance": {"_target_": "y"}}}

    class RecordingPipeline:
        def __init__(self, config_name, config_dir, dtype=torch.bfloat16, lazy_loading=False):
            seen["pipeline"] = (config_name, os.path.abspath(config_dir), dtype, lazy_loading)
            seen["denoiser_class"] = P.ActionMeshDenoiser
            P.load_config(config_name, config_dir)          # pipeline.py:66
            raise _Stop()

    monkeypatch.setattr(P, "load_config", fake_load_config)
    monkeypatch.setattr(P, "ActionMeshPipeline", RecordingPipeline)
    monkeypatch.chdir(tmp_path)
    argv = ["--backend", backend, "--reference-root", REF, "--", "--input", "clip.mp4", "--output_dir", str(tmp_path / "out"),
            "--dtype", "bfloat16", "--stage_1_steps", "50", *flags]
    with pytest.raises(_Stop):
        cli.main(argv)
    ref_dir = os.path.abspath(os.path.join(REF, "actionmesh", "configs"))
    assert seen["pipeline"] == (preset, ref_dir, torch.bfloat16, "--low_ram" in flags), "the script's own choices, untouched"
    from actionmesh_amd import HipDenoiser
    if backend == "hip":
        assert seen["load_config"][0] == preset.replace(".yaml", "_mi355x.yaml") and seen["load_config"][1] != ref_dir
        assert issubclass(seen["denoiser_class"], HipDenoiser) and dropin.is_installed()
    else:
        assert seen["load_config"] == (preset, ref_dir)
        assert not issubclass(seen["denoiser_class"], HipDenoiser) and not dropin.is_installed()
    assert (tmp_path / "out").is_dir()                      # the script's own mkdir ran (video_to_animated_mesh.py:219)


def test_cli_option_split_and_script_lookup():
    from actionmesh_amd import cli
    ours, rest = cli.split_args(["--backend", "reference", "--attn-dtype", "fp8", "--", "--input", "a", "--fast", "--seed", "3"])
    assert (ours.backend, ours.attn_dtype, ours.script) == ("reference", "fp8", "video_to_animated_mesh")
    assert rest == ["--input", "a", "--fast", "--seed", "3"]
    ours, rest = cli.split_args(["--input", "a", "--low_ram"])          # no `--`: unknown options fall through to the reference parser
    assert ours.backend == "hip" and rest == ["--input", "a", "--low_ram"]
    for name in cli.SCRIPTS:
        assert cli.find_script(name, REF) == os.path.join(REF, "inference", name + ".py")
    with pytest.raises(FileNotFoundError):
        cli.find_script("video_to_animated_mesh", "/nonexistent")
