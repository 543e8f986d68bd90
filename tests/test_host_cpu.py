"""CPU-only tests: the C-ABI library loads and exports every declared symbol, and the host-side
mirror of the reference interface (schedule, guidance, RoPE table, masked time, shard plan)
matches the reference's known answers.  No compute calls into the HIP library here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import actionmesh_amd as A
from actionmesh_amd import _lib, denoiser, scheduler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_library_loads_and_exports_every_declared_symbol(kind):
    """Both builds of the sources: libactionmesh_amd.so (bfloat16) and libactionmesh_amd_f16.so (-DAM_F16: `--dtype float16`)."""
    inc = os.path.join(ROOT, "include")
    header = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    declared = set(re.findall(r"\b(am_[a-z0-9_]+)\s*\(", header))
    declared -= {"am_status"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.lib(kind)        # raises if the .so is missing / lacks a symbol / ABI mismatch
    for name in declared:
        assert hasattr(lib, name)
    assert lib.am_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors vs the C header, measured by compiling a probe with gcc against include/."""
    import subprocess
    structs = {"am_config": _lib.AmConfig, "am_gemm_args": _lib.AmGemmArgs,
               "am_headpost_args": _lib.AmHeadPostArgs, "am_attn_args": _lib.AmAttnArgs, "am_nn_args": _lib.AmNnArgs,
               "am_peer_ring": _lib.AmPeerRing}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "actionmesh_amd.h"', '#include "actionmesh_amd_sharded.h"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _t in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0;}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = dict(l.split() for l in out if l)
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _t in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (error text via am_last_error)."""
    lib = _lib.lib()
    assert lib.am_gemm_bf16(None, None) != 0
    assert b"null" in lib.am_last_error()
    g = _lib.AmGemmArgs()
    g.M, g.N, g.K = 4, 8, 60
    assert lib.am_gemm_bf16(ctypes.byref(g), None) != 0
    assert b"multiple of 64" in lib.am_last_error()
    a = _lib.AmAttnArgs()
    assert lib.am_attention_bf16(ctypes.byref(a), None) != 0
    with pytest.raises(RuntimeError):
        _lib.check(-1, "x")


def test_no_cpu_fallback():
    m = A.HipDenoiser(num_layers=1, num_attention_heads=2, width=256, cross_attention_dim=64)
    m.load_state_dict({})
    with pytest.raises(RuntimeError):       # CPU device -> refused, not silently computed in torch
        m.forward(torch.zeros(2, 2, 4, 64), torch.zeros(2, 2, 3, 64), torch.zeros(2, 2), torch.zeros(2))
    with pytest.raises(TypeError):
        A.HipSchedulerFlow(num_inference_steps=1).denoise(torch.nn.Identity(), A.ClassifierFreeGuidance(),
                                                          torch.zeros(1, 2, 4, 64), torch.zeros(1, 2, 3, 64))


def test_schedule_kats(golden_dir):
    k = np.load(os.path.join(golden_dir, "kats.npz"))
    for n in (10, 15, 30, 50):
        t, d = A.HipSchedulerFlow(num_inference_steps=n, shift=3.0).get_schedule()
        assert np.array_equal(t.numpy(), k[f"sched_t_{n}"])
        assert np.array_equal(d.numpy(), k[f"sched_d_{n}"])
    s = A.HipSchedulerFlow(num_inference_steps=1)
    n = s.get_noise([8, 4], 1, 3, "cpu", torch.Generator().manual_seed(7))
    assert np.array_equal(n.numpy(), k["noise_seed7_small"])       # draw order: same, then independent


def test_guidance_matches_reference_kats(golden_dir):
    k = np.load(os.path.join(golden_dir, "kats.npz"))
    c = A.ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    assert np.allclose(c.aggregate_cfg(torch.tensor([[1.0], [2.0]])).numpy(), k["cfg_aggregate_1_2"])
    lat, ctx, m, f = c.cfg_at_inference(torch.ones(1, 2, 3, 4), torch.ones(1, 2, 5, 6), torch.tensor([[1.0, 0.0]]),
                                        torch.tensor([[0.0, 1.0]]))
    assert lat.shape[0] == 2 and bool((ctx[0] == 0).all()) and bool((ctx[1] == 1).all())
    assert torch.equal(m, torch.tensor([[1.0, 0.0], [1.0, 0.0]])) and f.shape == (2, 2)
    assert c.branches() == [[0, 1], [1, 1]]
    assert A.ClassifierFreeGuidance(False, [[0, 1], [1, 1]], [7.5]).branches() == [[1, 1]]
    with pytest.raises(AssertionError):
        A.ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [1.0, 2.0])


def test_rope_host_table_matches_reference_kat(golden_dir):
    k = np.load(os.path.join(golden_dir, "kats.npz"))
    cos, sin = denoiser.rope_tables_host(torch.arange(16.0)[None] + 5.0, 128)   # centred: +5 cancels
    assert cos.shape == (16, 64)
    assert np.allclose(cos.numpy(), k["rope_cos_128_16"][:, ::2], atol=1e-6)
    assert np.allclose(sin.numpy(), k["rope_sin_128_16"][:, ::2], atol=1e-6)


def test_masked_time_follows_reference_ordering():
    # temporal_denoiser.py:209-212: repeat(T) is b-fastest, the merged mask is (b t)
    t = denoiser.masked_time([700.0, 700.0], torch.tensor([[1.0, 0.0, 0.0], [1.0, 0.0, 0.0]]), 2, 3)
    assert t == [0.0, 700.0, 700.0, 0.0, 700.0, 700.0]
    t = denoiser.masked_time([1.0, 2.0], None, 2, 2)
    assert t == [1.0, 2.0, 1.0, 2.0]


def test_frame_shard_plan():
    p = A.FrameShardPlan(16, 4, 2)
    assert p.frames_local == 4 and p.frame_slice == slice(8, 12)
    x = torch.arange(2 * 16 * 3).view(2, 16, 3)
    assert torch.equal(p.slice_frames(x), x[:, 8:12])
    # a frame count the group does not divide (short windows: chunk_right / chunk_from on 5- or 7-frame videos) is not
    # sharded: every rank of the group computes all frames (replicas) instead of raising
    q = A.FrameShardPlan(16, 3, 2)
    assert q.replicated and (q.frame_world, q.frame_rank, q.frames_local, q.frame_slice) == (1, 0, 16, slice(0, 16))
    q = A.FrameShardPlan(7, 8, 5, batch=2, cfg_groups=2)
    assert q.replicated and (q.group_size, q.frame_world, q.cfg_rank, q.frame_rank, q.frames_local) == (4, 1, 1, 0, 7)
    assert q.local_times(list(range(14))) == list(range(7, 14))
    with pytest.raises(ValueError):
        A.FrameShardPlan(16, 4, 4)
    # CFG-parallel x frame shards: rank = cfg_rank * frame_world + frame_rank
    p = A.FrameShardPlan(16, 8, 6, batch=2, cfg_groups=2)
    assert (p.frame_world, p.frame_rank, p.cfg_rank, p.frames_local, p.batch_local) == (4, 2, 1, 4, 1)
    assert p.frame_group_ranks(1) == [4, 5, 6, 7]
    x = torch.arange(2 * 16 * 3).view(2, 16, 3)
    assert torch.equal(p.slice_local(x), x[1:2, 8:12])
    assert p.local_times(list(range(32))) == [16 + 8, 16 + 9, 16 + 10, 16 + 11]
    p2 = A.FrameShardPlan(16, 2, 1, batch=2, cfg_groups=2)
    assert p2.frame_world == 1 and p2.frames_local == 16 and p2.batch_slice == slice(1, 2)
    with pytest.raises(ValueError):
        A.FrameShardPlan(16, 4, 0, batch=3, cfg_groups=2)


def test_perm16_is_an_involution_matching_the_mfma_layout():
    from actionmesh_amd import ops
    idx = ops.perm16_index(64)
    assert torch.equal(idx[idx], torch.arange(64))
    # k-slot (hi, j) of the P.V MFMA B operand carries key (j&3) + 8*(j>>2) + 4*hi (32x32 C/D layout)
    for hi in range(2):
        for j in range(8):
            assert int(idx[hi * 8 + j]) == (j & 3) + 8 * (j >> 2) + 4 * hi


def test_attn64_register_audit(tmp_path):
    """The 4x64 attention kernel names AccVGPRs a[64:255] literally in inline asm; the ISA audit that makes this sound
    (actionmesh_amd/csrc/audit_attn64.py: no scratch, no compiler access to a[64:255], no compiler-generated touch of an
    in-flight QK^T MFMA result) is a Makefile build step; this test runs the same script on a fresh assembly and checks
    that the script does reject a violation."""
    import importlib.util
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "actionmesh_amd", "csrc")
    out = tmp_path / "a64.s"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mno-amdgpu-ieee", "-fno-honor-nans",
             "--cuda-device-only", "-S", "-o", str(out), os.path.join(csrc, "am_attention64.hip")]
    subprocess.run([hipcc] + flags, check=True, capture_output=True, timeout=600)
    spec = importlib.util.spec_from_file_location("audit_attn64", os.path.join(csrc, "audit_attn64.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = out.read_text()
    errs, n = mod.audit(text)
    assert n >= 4 and not errs, errs[:5]
    bad = text.replace("#ASMEND", "#ASMEND\n\tv_accvgpr_read_b32 v1, a100", 1)
    errs, _ = mod.audit(bad)
    assert errs and "asm-owned AccVGPR" in errs[0]
    # placement guard (round 2): hipcc once hoisted the softmax steps above the QK^T MFMAs - every other gap empty, the next one
    # twice as full.  tools/attn64_gaps.py counts what the build left in each MFMA gap of the product (lazy) kernel's main loop.
    spec = importlib.util.spec_from_file_location("attn64_gaps", os.path.join(root, "tools", "attn64_gaps.py"))
    gaps_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gaps_mod)
    run = gaps_mod.gaps(gaps_mod.kernel_body(text, "ILi8ELi0ELb0ELi0ELb1E"))
    slots, n_mfma, started = [], 0, False
    for kind, v in run:
        if kind == "gap" and v.get("bar") and not started and n_mfma > 40:
            started = True
            continue
        if kind == "mfma":
            n_mfma += 1
        if started and kind == "gap":
            slots.append((v.get("valu", 0) + 2 * v.get("exp", 0), v.get("dma", 0)))
        if started and len(slots) >= 64:
            break
    assert len(slots) == 64
    body_slots = [s_ for s_, dma in slots[:63] if not dma and s_ < 50]            # steady gaps of one tile (not the phase boundary)
    assert len(body_slots) >= 50 and min(body_slots) >= 1 and max(body_slots) <= 8, body_slots      # spread: no empty gap, no doubled one
    assert "am_attention64.audit" in open(os.path.join(csrc, "Makefile")).read()



@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_no_swapped_packed_f32_in_the_built_library(tmp_path, kind):
    """Round 3 root cause of the same-device divergence: on MI355X `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[0,0]`
    (low result = src0.lo x src1.HI) returns a wrong low half in lanes 48-63 while another process runs bf16 GEMMs on the device
    (tools/repro/pk_mul_cross_process.hip).  hipcc's SLP vectoriser emitted it for head_post's RoPE rotation; the row-wise kernels are
    now built without SLP.  This audit disassembles the device code of the BUILT library: no packed-FP32 instruction whose low half
    reads the high half of an operand, anywhere; and no packed-FP32 instruction at all in the row-wise kernels."""
    import glob
    import re
    import shutil
    import subprocess
    from actionmesh_amd import _lib as L_
    LIB_PATH = L_.LIB_PATH if kind == "bf16" else L_.LIB_PATH_F16
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(LIB_PATH) and os.path.exists(objdump)):
        pytest.skip("library or llvm-objdump missing")
    shutil.copy(LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(glob.glob(str(tmp_path / "lib.so.*gfx950")))
    assert objs, "no gfx950 code objects in the library"
    swapped = re.compile(r"v_pk_(mul|fma|add)_f32 .*op_sel:\[(0,1|1,0|0,1,[01]|1,0,[01]|0,0,1|1,1,0)")
    n_pk, kernel = 0, None
    for o in objs:
        for line in subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True).stdout.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                kernel = m.group(1)
                continue
            if "v_pk_" in line and "_f32" in line:
                n_pk += 1
                assert not swapped.search(line), f"{kernel}: {line.strip()}"
                assert not any(k in (kernel or "") for k in ("head_post_kernel", "layernorm_kernel", "flow_step", "f32_to_bf16")), \
                    f"row-wise kernel {kernel} contains a packed-FP32 instruction: {line.strip()}"
    assert n_pk > 100, "the audit did not see the GEMM / attention kernels' packed instructions - wrong objects?"


def test_autocast_kind_follows_the_callers_region(monkeypatch):
    """_lib.autocast_kind: the pinned kind wins; otherwise float16 only inside an enabled cuda autocast region whose dtype is float16
    (the reference pipeline's `--dtype float16`), bfloat16 everywhere else; and on a torch without get_autocast_dtype the per-device
    getters are used instead of silently answering bfloat16."""
    import torch
    assert _lib.autocast_kind("f16") == "f16" and _lib.autocast_kind("bf16") == "bf16"
    assert _lib.autocast_kind(None) == "bf16"                                   # no autocast region here
    monkeypatch.setattr(torch, "is_autocast_enabled", lambda *a: True)
    monkeypatch.setattr(torch, "get_autocast_dtype", lambda *a: torch.float16)
    assert _lib.autocast_kind(None) == "f16"
    monkeypatch.setattr(torch, "get_autocast_dtype", lambda *a: torch.bfloat16)
    assert _lib.autocast_kind(None) == "bf16"
    monkeypatch.delattr(torch, "get_autocast_dtype")                            # an older torch
    monkeypatch.setattr(torch, "get_autocast_gpu_dtype", lambda: torch.float16)
    assert _lib.autocast_kind(None) == "f16"
    assert _lib.kind_of(torch.float16) == "f16" and _lib.kind_of("bfloat16") == "bf16"
    with pytest.raises(ValueError, match="bfloat16 or float16"):
        _lib.kind_of(torch.float32)
