#!/usr/bin/env python3
"""Build check for the hand-issued LDS-DMA blocks (global_load_lds_dwordx4 behind `s_mov_b32 m0, ...` in inline asm:
approx_kernels.hpp score_approx32_kernel's GL = 1 gather, encoder_kernels.hpp gemm_planes_kernel / gemm_planes2_kernel).

hipcc does not accept M0 in an asm clobber list (a reserved register), so those blocks write M0 without telling the
compiler.  That is safe only while NOTHING ELSE in the same kernel keeps a value in M0 -- a builtin that lowers to
s_movrel / v_movrel (indirect register indexing), ds_*_gds, s_sendmsg, v_interp, or the compiler's own
`global_load_lds` lowering would.  This tool disassembles every gfx950 code object of libcolbert_hip.so and, for every
kernel that issues an LDS-DMA, fails if any instruction touches M0 other than

    s_mov_b32 m0, <sgpr|imm>            (the asm block's own set-up)
    s_add_u32 / s_add_i32 m0, m0, <imm> (the second DMA of a block: the next 1 KB of the LDS ring)

and if any global_load_lds* is not preceded, since the previous global_load_lds* / branch target, by one of those writes.
It also fails when a kernel WITHOUT an LDS-DMA touches M0 at all in a file that has such kernels (a sign that the compiler
started using M0 itself).  Exit code 0 = clean.

    python tools/check_m0.py [path/to/libcolbert_hip.so]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALLOWED = (re.compile(r"^s_mov_b32 m0, (s\d+|0x[0-9a-f]+|\d+|-?\d+)$"),
           re.compile(r"^s_add_(u|i)32 m0, m0, (0x[0-9a-f]+|\d+)$"))
# instructions that read M0 implicitly on gfx9-class targets (besides the LDS-DMA itself)
IMPLICIT_M0 = re.compile(r"^(s_movrel|v_movrel|s_sendmsg|v_interp|ds_\w+_gds|ds_gws|s_ttrace|v_readlane_b32 \S+, \S+, m0|v_writelane_b32 \S+, \S+, m0)")


def code_objects(so_path):
    data = open(so_path, "rb").read()
    out, i = [], 0
    while True:
        j = data.find(b"__CLANG_OFFLOAD_BUNDLE__", i)
        if j < 0:
            return out
        (num,) = struct.unpack_from("<Q", data, j + 24)
        o = j + 32
        for _ in range(num):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                out.append(data[j + off:j + off + size])
        i = j + 24


def check(so_path, verbose=True):
    problems, kernels_with_dma, dma_count = [], 0, 0
    objs = code_objects(so_path)
    if not objs:
        return ["no gfx950 code object found in " + so_path], 0, 0
    for blob in objs:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        kernels, name, body = [], None, []
        for line in text.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
            if m:
                if name is not None:
                    kernels.append((name, body))
                name, body = m.group(1), []
                continue
            ins = line.split("//")[0].strip()
            if ins and name is not None:
                body.append(ins)
        if name is not None:
            kernels.append((name, body))
        for name, body in kernels:
            dma = [k for k, ins in enumerate(body) if ins.startswith("global_load_lds")]
            touch = [k for k, ins in enumerate(body) if re.search(r"\bm0\b", ins) or IMPLICIT_M0.match(ins)]
            if not dma:
                if touch:
                    problems.append(f"{name}: touches M0 without an LDS-DMA: {body[touch[0]]}")
                continue
            kernels_with_dma += 1
            dma_count += len(dma)
            for k in touch:
                ins = body[k]
                if not any(p.match(ins) for p in ALLOWED):
                    problems.append(f"{name}: instruction outside the asm blocks touches M0: {ins}")
            # every DMA is fed by an M0 write of the block it belongs to: walking back from it, the first M0-touching
            # instruction must be an allowed write and no label / branch may lie in between (the blocks are straight-line)
            for k in dma:
                j = k - 1
                ok = False
                while j >= 0:
                    ins = body[j]
                    if any(p.match(ins) for p in ALLOWED):
                        ok = True
                        break
                    if ins.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "global_load_lds")) or ins.endswith(":"):
                        break
                    j -= 1
                if not ok:
                    problems.append(f"{name}: {body[k]} is not preceded by its own M0 write")
    if verbose:
        print(f"{os.path.basename(so_path)}: {len(objs)} code objects, {kernels_with_dma} kernels issue {dma_count} LDS-DMAs, "
              f"{len(problems)} problem(s)")
        for p in problems[:20]:
            print("  " + p)
    return problems, kernels_with_dma, dma_count


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "colbert.jl_amd", "csrc", "libcolbert_hip.so")
    sys.exit(1 if check(so)[0] else 0)
