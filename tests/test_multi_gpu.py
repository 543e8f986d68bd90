"""Real multi-rank runs (one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI).  Auto-skip on boxes with
fewer than two devices (the build pool hands out single-GPU boxes; a multi-GPU driver box exercises these): RCCL ranks,
the CFG-branch x frame-shard groups, the asynchronous [K | V^T] all-gather overlapped with the local-shard attention, and
1-vs-N parity of the sharded forward (tools/mgpu_selftest.py: rel-L2 < 1e-2 vs the unsharded forward on rank 0)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, extra=(), tool="mgpu_selftest", env_extra=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", f"{tool}.py"), *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and "AssertionError" not in r.stderr and "RuntimeError" not in r.stderr:
        # a rendezvous / launcher hiccup (port reuse between back-to-back torchruns), not a verdict of the tool: once more
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and f"[{tool}] ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_forward_on_real_ranks(world):
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, found {torch.cuda.device_count() if torch.cuda.is_available() else 0}")
    out = _run(world)
    print(out.strip().splitlines()[-2])


def test_short_window_falls_back_to_replicas():
    """7 frames over 2 frame shards per CFG branch do not divide: every rank of a branch computes all frames."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 4:
        pytest.skip("needs 4 GPUs")
    _run(4, ("--frames", "7"))


@pytest.mark.parametrize("world", [2, 4])
def test_copy_engine_exchange_across_processes_on_one_device(world):
    """The copy-engine exchange back-end (sharding.PeerExchange; ACTIONMESH_AMD_EXCHANGE=peer): `world` processes share ONE
    GPU, so this runs on the single-GPU boxes too - IPC-mapped gather buffers, SDMA pushes and the arrived / consumed flag
    protocol between real processes, three forwards in a row, against the unsharded forward."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _run(world, ("--same-device",), tool="peer_selftest")
    print(out.strip().splitlines()[-2])


@pytest.mark.parametrize("world", [2, 4])
def test_fp8_shards_across_processes_on_one_device(world):
    """attn_dtype = fp8 under frame sharding between real processes (one device): the QUANTISED shards travel through the copy-engine
    exchange, every rank's self-attention runs in fp8 and never in bf16 (am_attention_counters), repeated forwards are bit-identical."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _run(world, ("--same-device", "--dtype", "fp8"), tool="peer_selftest")
    print(out.strip().splitlines()[-2])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_fp8_rccl_ranks(world):
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    _run(world, ("--dtype", "fp8"))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_copy_engine_exchange_on_real_ranks(world):
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    _run(world, tool="peer_selftest")
    _run(world, env_extra={"ACTIONMESH_AMD_EXCHANGE": "peer"})          # the same back-end under HipDenoiser + RCCL control plane
