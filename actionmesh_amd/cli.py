"""`python -m actionmesh_amd.cli [--backend {hip,reference}] [--attn-dtype {bf16,fp8,fp8_fast}] [--stage2-hip] [--script NAME]
                                 [--reference-root DIR] -- <the reference CLI's own arguments>`

Runs the reference's UNMODIFIED command-line script (inference/video_to_animated_mesh.py:120-248, or
inference/video_and_3d_to_animated_mesh.py with `--script video_and_3d_to_animated_mesh`) with its own argument parser and
its own `--fast / --low_ram / --dtype / --stage_1_steps ...` flags, after `actionmesh_amd.install()` has put the MI355X
sampler and denoiser behind `ActionMeshPipeline` (actionmesh_amd/dropin.py): the preset the script derives from `--fast` /
`--low_ram` (lines 199-210) is served as its `_mi355x` overlay, and `ActionMeshDenoiser.from_pretrained` (pipeline.py:180)
builds a HipDenoiser.  `--backend reference` runs the same script untouched (`--backend {reference,hip}`).
The reference script keeps its code under `if __name__ == "__main__":`, so it is executed with runpy as `__main__`.
"""
from __future__ import annotations

import argparse
import os
import runpy
import sys
from typing import List, Optional, Tuple

SCRIPTS = ("video_to_animated_mesh", "video_and_3d_to_animated_mesh")


def find_script(name: str, reference_root: Optional[str]) -> str:
    roots = [reference_root, os.environ.get("ACTIONMESH_ROOT")]
    if not any(roots):
        import actionmesh          # the installed / checked-out reference: <root>/actionmesh/__init__.py
        roots.append(os.path.dirname(os.path.dirname(os.path.abspath(actionmesh.__file__))))
    for r in roots:
        if r:
            path = os.path.join(r, "inference", name + ".py")
            if os.path.isfile(path):
                return path
    raise FileNotFoundError(f"actionmesh_amd.cli: inference/{name}.py not found under {[r for r in roots if r]} "
                            "(pass --reference-root or set ACTIONMESH_ROOT)")


def split_args(argv: List[str]) -> Tuple[argparse.Namespace, List[str]]:
    """Our own options come first; everything else (after an optional `--`) is handed to the reference parser verbatim."""
    ap = argparse.ArgumentParser(prog="python -m actionmesh_amd.cli", add_help=False,
                                 description="Run the reference ActionMesh CLI on the MI355X backend (or untouched).")
    ap.add_argument("--backend", choices=["hip", "reference"], default="hip")
    ap.add_argument("--attn-dtype", choices=["bf16", "fp8", "fp8_fast"], default="bf16",
                    help="inflated self-attention arithmetic: bf16 (default), e4m3 MFMA (fp8), or its exponent-field form (fp8_fast)")
    ap.add_argument("--stage2-hip", action="store_true", help="also decode Stage II on the HIP kernels (HipAutoencoder)")
    ap.add_argument("--script", choices=SCRIPTS, default=SCRIPTS[0])
    ap.add_argument("--reference-root", default=None)
    ap.add_argument("--amd-help", action="store_true", help="this wrapper's options (plain --help shows the reference CLI's)")
    ours, rest = ap.parse_known_args(argv)
    if rest and rest[0] == "--":
        rest = rest[1:]
    if ours.amd_help:
        ap.print_help()
        raise SystemExit(0)
    return ours, rest


def main(argv: Optional[List[str]] = None) -> None:
    ours, rest = split_args(list(sys.argv[1:] if argv is None else argv))
    if ours.reference_root:
        sys.path.insert(0, ours.reference_root)
    script = find_script(ours.script, ours.reference_root)
    if ours.backend == "hip":
        from . import dropin
        dropin.install(attn_dtype=ours.attn_dtype, stage2=ours.stage2_hip)
    old_argv = sys.argv
    sys.argv = [script] + rest
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        sys.argv = old_argv


if __name__ == "__main__":
    main()
