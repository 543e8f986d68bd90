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


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_phase_loop_in_c_equals_the_python_loop(dtype):
    """the sharded forward's per-layer phase loop as ONE C call (am_forward_sharded_peer, include/actionmesh_amd_sharded.h:
    begin, pre, pushes + local attention, wait, post, consumed, end) against sharding.sharded_forward's Python loop - two processes on
    one device, four forwards alternating between the two drivers on the SAME exchange ring (the sequence flags keep turning
    across them), outputs bit-identical on every rank, and the usual comparison against the unsharded forward."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _run(2, ("--same-device", "--loop", "both", "--forwards", "4", "--dtype", dtype), tool="peer_selftest")
    assert out.count("bit-identical") == 2, out[-2000:]


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


def _bench(world, extra=(), bare=False):
    """bench.py itself, as the driver launches it (torch.distributed.run for N > 1), with every rank on ONE device.  `bare`: N > 1
    WITHOUT a launcher around it - bench.py re-executes itself under torch.distributed.run (round 6)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--shape", "small", "--no-cpu-baseline", *extra]
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args]
    elif bare:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args, "--same-device"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), *args, "--same-device"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly ONE JSON line on rank 0, got {len(lines)}:\n" + r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_bench_world_gt_1_branch_runs_on_one_device(dtype):
    """bench.py's N > 1 branch (process-group init, the CFG x frame sub-groups inside HipDenoiser, the
    barrier-bracketed timing, the MAX all-reduce over ranks, the one JSON line with n_gpus / parallelism / roofline per rank count) had
    never executed anywhere.  `--same-device` runs exactly that code with all ranks on cuda:0 (gloo control plane, copy-engine exchange):
    worlds 2 (pure CFG split) and 4 (CFG x 2 frame shards, one K/V exchange per layer), against the 1-rank run of the same command.
    Value-independent invariants only - the timing of ranks sharing a device means nothing."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    extra = ("--dtype", dtype)
    one = _bench(1, extra)
    assert one["n_gpus"] == 1 and one["config"]["parallelism"] == "single GPU" and "same_device_dry_run" not in one
    fp1 = one["latents_fingerprint"]
    launches1 = one["attention_launches"][dtype]
    for world in (2, 4):
        d = _bench(world, extra)
        assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["same_device_dry_run"] is True
        assert d["metric"].startswith("DRY RUN") and d["scaling"] == "strong" and d["dtype"] == dtype and d["value"] > 0
        assert d["config"]["parallelism"] == f"cfg-branch x2 * frame-shard x{world // 2}"
        assert d["exchange_backend"].startswith("peer")
        # rank 0 runs ONE CFG branch: every layer of every forward is one self-attention launch sequence on it; with frame shards
        # (world 4) a layer is two phases (local shard, then the remote ones) but still counted per layer by the engine
        other = "bf16" if dtype == "fp8" else "fp8"
        assert d["attention_launches"][other] == 0 and d["attention_launches"][dtype] > 0
        assert d["attention_launches"][dtype] in (launches1, 2 * launches1), (d["attention_launches"], launches1)
        rf = d["roofline"]
        assert rf["bound"] == "mfma" and rf["achieved"] > 0 and 0 < rf["frac"] < 1 and rf["launch_ms"] > 0
        fp = d["latents_fingerprint"]
        assert fp["after_steps"] == fp1["after_steps"]
        tol = 5e-2 if dtype == "fp8" else 2e-2          # the sharding tolerance of tools/peer_selftest.py, on O(1) latents
        assert abs(fp["rms"] - fp1["rms"]) < tol * fp1["rms"], (fp, fp1)
        assert max(abs(a - b) for a, b in zip(fp["sample"], fp1["sample"])) < tol * max(1.0, fp1["rms"]), (fp, fp1)
        # round 5: the line checks ITSELF - rank 0 re-ran the same steps on an unsharded engine and compared the final latents
        assert d["fingerprint_ok"] is True and d["fingerprint_check"]["legs"]["peer"] <= 3e-2, d["fingerprint_check"]
        assert d["exchange_ab"]["peer"]["ok"] is True and d["exchange_ab"]["peer"]["ms_per_step"] > 0
        if world == 4:
            assert isinstance(d["exchange_ab"]["peer"]["flags_fine_grained"], bool)


def test_bench_exchange_ab_falls_back_when_one_backend_fails():
    """`--exchange ab` (the default of a real multi-GPU launch) with every rank on ONE device: the RCCL leg cannot run here and raises -
    the line must still come out, carry the copy-engine leg as `value`, and say what happened to the other one (VERDICT r04 next #2:
    "falling back rather than dying if one fails")."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _bench(4, ("--exchange", "ab"))
    ab = d["exchange_ab"]
    assert ab["rccl"]["ok"] is False and "RCCL refuses two ranks" in ab["rccl"]["error"]
    assert ab["peer"]["ok"] is True and d["exchange_backend"].startswith("peer") and d["value"] > 0
    assert d["fingerprint_ok"] is True


def test_bench_pure_frame_sharding():
    """`--cfg-parallel 0`: north_star's partition - every rank computes BOTH guidance branches of T / N frames (4 ranks x 1 frame of
    the 4-frame plumbing shape), one K/V exchange per layer among all ranks."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _bench(4, ("--cfg-parallel", "0"))
    assert d["config"]["parallelism"] == "cfg-branch x1 * frame-shard x4"
    assert d["fingerprint_ok"] is True and d["exchange_ab"]["peer"]["ok"] is True


@pytest.mark.parametrize("world", [2, 4])
def test_bench_launches_itself_and_preflights(world):
    """`python bench.py --gpus N` with NO launcher around it re-executes itself under torch.distributed.run, the
    ONE JSON line is the last line of stdout and the exit code is 0; in front of the legs every rank ran the seconds-long pre-flight of
    the exchange back-end in a killable child process and the verdict is in `exchange_ab.preflight` (sharded vs single-rank latents of
    the pre-flight problem within the stated 3e-2)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _bench(world, bare=True)
    assert d["launcher"]["self_launched"] is True and d["launcher"]["attempts"][0]["returncode"] == 0
    assert d["launcher"]["attempts"][0]["argv"][:2] == ["--gpus", str(world)] and "torch.distributed.run" in d["launcher"]["command"]
    assert d["n_gpus"] == world and d["value"] > 0 and d["fingerprint_ok"] is True
    pf = d["exchange_ab"]["preflight"]["peer"]
    assert pf["ok"] is True and pf["rel_l2_vs_single_rank"] <= 3e-2 and pf["seconds"] < 90, pf
    assert d["exchange_ab"]["peer"]["ok"] is True


def test_bench_preflight_keeps_a_dead_backend_from_the_legs():
    """`--exchange ab` with every rank on one device: RCCL cannot run there - the pre-flight says so in seconds, no leg is spent on it,
    the copy-engine leg is the result."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _bench(4, ("--exchange", "ab"))
    ab = d["exchange_ab"]
    assert ab["preflight"]["rccl"]["ok"] is False and ab["preflight"]["peer"]["ok"] is True
    assert ab["rccl"]["ok"] is False and ab["rccl"].get("skipped") is True and "RCCL refuses two ranks" in ab["rccl"]["error"]
    assert ab["peer"]["ok"] is True and d["value"] > 0
    # and with the pre-flight off the leg itself fails and the fallback is what carries the line (the round-5 behaviour)
    d = _bench(4, ("--exchange", "ab", "--no-preflight"))
    assert d["exchange_ab"]["rccl"]["ok"] is False and "skipped" not in d["exchange_ab"]["rccl"] and d["exchange_ab"]["peer"]["ok"] is True
