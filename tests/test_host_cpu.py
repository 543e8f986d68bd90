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


def test_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "actionmesh_amd.h")).read()
    declared = set(re.findall(r"\b(am_[a-z0-9_]+)\s*\(", header))
    declared -= {"am_status"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.lib()            # raises if the .so is missing / lacks a symbol / ABI mismatch
    for name in declared:
        assert hasattr(lib, name)
    assert lib.am_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors vs the C header, measured by compiling a probe with gcc against include/."""
    import subprocess
    structs = {"am_config": _lib.AmConfig, "am_gemm_args": _lib.AmGemmArgs,
               "am_headpost_args": _lib.AmHeadPostArgs, "am_attn_args": _lib.AmAttnArgs}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "actionmesh_amd.h"', 'int main(void){']
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
    with pytest.raises(ValueError):
        A.FrameShardPlan(16, 3, 0)
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
    """The 4x64 attention kernel names AccVGPRs a[64:255] literally in inline asm (O accumulators, Q fragments).
    That is only sound if hipcc itself never touches a[64:255] in that kernel: no scratch spills, and every
    compiler-generated AccVGPR access (when the kernel needs more than 256 arch VGPRs the allocator parks values in
    AccVGPRs, lowest free first, whatever the asm clobber lists say) inside a[0:63], which the asm leaves alone
    (tools/gen_attn64_asm.py; cdna guide 'keep out of registers you name')."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "actionmesh_amd", "csrc", "am_attention64.hip")
    out = tmp_path / "a64.s"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mno-amdgpu-ieee", "-fno-honor-nans",
             "--cuda-device-only", "-S", "-o", str(out), src]
    subprocess.run([hipcc] + flags, check=True, capture_output=True, timeout=600)
    text = out.read_text()
    kernels = re.findall(r"\.agpr_count:\s+(\d+)\n\s+\.name:\s+(\S*attn_fwd64_kernel\S*)\n\s+\.private_segment_fixed_size:\s+(\d+)", text)
    if not kernels:   # field order differs between compiler versions: fall back to independent searches
        names = re.findall(r"\.name:\s+(\S*attn_fwd64_kernel\S*)", text)
        assert names, "no attn_fwd64_kernel in the assembly"
        kernels = [(a, n, p) for a, n, p in zip(re.findall(r"\.agpr_count:\s+(\d+)", text), names,
                                                re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text))]
    for agprs, name, scratch in kernels:
        assert int(agprs) == 256, f"{name}: {agprs} AccVGPRs allocated, the asm owns a[64:255]"
        assert int(scratch) == 0, f"{name}: spills to scratch ({scratch} B)"
    # hipcc may park values in AccVGPRs of its own (a0..a63) - never in the asm-owned range
    in_asm = False
    for ln in text.splitlines():
        code = ln.split(";")[0]
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        elif not in_asm and re.match(r"\s+(v_|ds_|global_|buffer_|scratch_|flat_)", code):
            for m in re.finditer(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", code):
                lo = int(m.group(1) if m.group(1) is not None else m.group(3))
                hi_ = int(m.group(2)) if m.group(2) is not None else lo
                assert hi_ < 64, f"compiler-generated access to an asm-owned AccVGPR: {ln.strip()}"
    # hipcc does not know the asm statements are MFMAs: nothing it generates (copies, parking in AccVGPRs, softmax steps)
    # may touch the arch-VGPR destination of a QK^T MFMA within the 11 wait states an 8-pass MFMA needs (14 checked)
    def vregs(tok):
        out = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1) if m.group(1) is not None else [int(m.group(3))])
        return out
    in_asm, hot = False, {}
    for ln in text.splitlines():
        code = ln.split(";")[0].rstrip()
        if "#ASMSTART" in ln:
            in_asm = True
            continue
        if "#ASMEND" in ln:
            in_asm = False
            continue
        m = re.match(r"\s+([a-z]\S*)\s*(.*)", code)
        if not m:
            continue
        op, args = m.group(1), m.group(2)
        step = int(args.strip()) + 1 if op == "s_nop" else 1
        if not in_asm and op != "s_nop":
            touched = vregs(args) & set(hot)
            assert not touched, f"compiler-generated access to an in-flight MFMA result: {ln.strip()}"
        hot = {k: v - step for k, v in hot.items() if v - step > 0}
        if in_asm and op.startswith("v_mfma") and args.split(",")[0].strip().startswith("v"):
            hot.update({r: 14 for r in vregs(args.split(",")[0])})


def test_gemm4w_register_audit(tmp_path):
    """The 4-wave GEMM main loop (am_gemm4w.hip) names all 256 AccVGPRs literally in inline asm: hipcc must not generate
    a single AccVGPR access of its own in that kernel, and must not spill."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "actionmesh_amd", "csrc", "am_gemm4w.hip")
    out = tmp_path / "g4.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-I", os.path.join(root, "include"),
                    "-S", "-o", str(out), src], check=True, capture_output=True, timeout=600)
    text = out.read_text()
    assert re.search(r"\.agpr_count:\s+256", text) and re.search(r"\.private_segment_fixed_size:\s+0\b", text)
    in_asm, n_mfma = False, 0
    for ln in text.splitlines():
        code = ln.split(";")[0]
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        elif in_asm and "v_mfma" in code:
            n_mfma += 1
        elif not in_asm and re.match(r"\s+(v_|ds_|global_|buffer_|scratch_|flat_)", code):
            assert not re.search(r"\ba\[\d+:\d+\]|\ba\d+\b", code), f"compiler-generated AccVGPR access: {ln.strip()}"
    assert n_mfma >= 32

