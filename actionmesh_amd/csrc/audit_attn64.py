#!/usr/bin/env python3
"""ISA audit of the 4x64 attention kernel (am_attention64.hip).  Build step: the Makefile assembles the kernel to
`am_attention64.s` with the flags of the object file and runs this script on it; a violation FAILS THE BUILD
(the audit used to live only in a test a build could skip).

The kernel names AccVGPRs a[64:255] literally in inline asm (O accumulators, pre-scaled Q fragments).  That is only sound
if hipcc itself never touches a[64:255] in that kernel:
  1. every attn_fwd64_kernel instantiation allocates all 256 AccVGPRs and spills nothing to scratch;
  2. every compiler-generated AccVGPR access (beyond 256 arch VGPRs the allocator parks values in AccVGPRs, lowest free
     first, whatever the asm clobber lists say) stays inside a[0:63], which the asm leaves alone (tools/gen_attn64_asm.py);
  3. nothing hipcc generates touches the arch-VGPR destination of a QK^T MFMA within the wait states an 8-pass MFMA needs
     (11; 14 checked) - hipcc does not know the asm statements are MFMAs and pads nothing around them.
Usage: audit_attn64.py <file.s>      exit status 0 = clean
"""
import re
import sys


def audit(text: str):
    errors = []
    kernels = re.findall(
        r"\.agpr_count:\s+(\d+)\n\s+\.name:\s+(\S*attn_fwd64_kernel\S*)\n\s+\.private_segment_fixed_size:\s+(\d+)", text)
    if not kernels:   # field order differs between compiler versions: fall back to independent searches
        names = re.findall(r"\.name:\s+(\S*attn_fwd64_kernel\S*)", text)
        if not names:
            return ["no attn_fwd64_kernel in the assembly"]
        kernels = list(zip(re.findall(r"\.agpr_count:\s+(\d+)", text), names,
                           re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)))
    for agprs, name, scratch in kernels:
        if int(agprs) != 256:
            errors.append(f"{name}: {agprs} AccVGPRs allocated, the asm owns a[64:255]")
        if int(scratch) != 0:
            errors.append(f"{name}: spills to scratch ({scratch} B)")

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
        if not in_asm and re.match(r"(v_|ds_|global_|buffer_|scratch_|flat_)", op):
            for a in re.finditer(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", args):
                lo = int(a.group(1) if a.group(1) is not None else a.group(3))
                hi = int(a.group(2)) if a.group(2) is not None else lo
                if hi >= 64:
                    errors.append(f"compiler-generated access to an asm-owned AccVGPR: {ln.strip()}")
        step = int(args.strip()) + 1 if op == "s_nop" else 1
        if not in_asm and op != "s_nop":
            if vregs(args) & set(hot):
                errors.append(f"compiler-generated access to an in-flight MFMA result: {ln.strip()}")
        hot = {k: v - step for k, v in hot.items() if v - step > 0}
        if in_asm and op.startswith("v_mfma") and args.split(",")[0].strip().startswith("v"):
            hot.update({r: 14 for r in vregs(args.split(",")[0])})
    return errors, len(kernels)


if __name__ == "__main__":
    res = audit(open(sys.argv[1]).read())
    errs, n = res if isinstance(res, tuple) else (res, 0)
    for e in errs[:20]:
        print("audit_attn64: " + e, file=sys.stderr)
    if errs:
        sys.exit(1)
    print(f"audit_attn64: {n} attn_fwd64_kernel instantiations clean")
