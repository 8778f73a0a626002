#!/usr/bin/env python3
"""Build-time audit of the gfx950 code objects (run by csrc/Makefile after the objects are built).

Fails when a kernel contains a packed-fp32 ARITHMETIC instruction (v_pk_add/mul/fma_f32) whose `op_sel:[...]` makes a
LOW result lane read the HIGH dword of a source pair. hipcc (ROCm 7.2) emits that form when it packs the scalars of
neighbouring rows into one 64-bit register and broadcasts the odd one; on MI355X every build of the LayerNorm-backward
GEMM epilogue that contained it gave intermittently wrong rows in the passes that used it. The
sources avoid it (XF_PIN_SCALAR in gemm.hip, -fno-slp-vectorize for the two cold files); this check keeps a compiler
or source change from bringing it back unnoticed.

Second rule (round 4): the kernels that carry the LayerNorm-backward epilogue (gemm_kernel<..., EPI_DX_LNBWD = 6, ...>,
ffn_bwd_dx_fused_kernel) must not store to LDS straight from the MFMA accumulator file (`ds_write_b32 v, aN`). Every failing
diagnostic build did (launch bound 1: the accumulators live in AGPRs); the product build (launch bound 3) keeps its
accumulators in VGPRs and the epilogue routes each value through a VGPR anyway -- the second resource the failing builds
share beyond the op_sel marker. Cheap to hold, so it is held.

usage: check_isa.py obj1.o obj2.o ...   (prints a per-object count; exit 1 on any hit)"""
import pathlib
import re
import subprocess
import sys
import tempfile

LLVM = pathlib.Path("/opt/rocm/lib/llvm/bin")
BAD = re.compile(r"\bv_pk_(add|mul|fma)_f32\b.*\bop_sel:\[")
AGPR_LDS_STORE = re.compile(r"\bds_write_b(32|64|128)\s+v\d+, a\[?\d+")
LNBWD_KERNEL = re.compile(r"ffn_bwd_dx_fused_kernel|gemm_kernelI\w+?Lb[01]ELb[01]ELi6E")


def disassemble(obj: pathlib.Path) -> str:
    with tempfile.TemporaryDirectory() as td:
        local = pathlib.Path(td) / obj.name
        local.write_bytes(obj.read_bytes())
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", local.name], cwd=td, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = sorted(pathlib.Path(td).glob(local.name + ".*amdgcn*gfx950*"))
        if not cos:  # a host-only translation unit (runtime.hip: streams / events / copies, no kernel)
            return ""
        return subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(cos[0])], check=True, capture_output=True,
                              text=True).stdout


def main(argv):
    bad_total = 0
    for name in argv:
        text = disassemble(pathlib.Path(name))
        kernel, hits = "?", {}
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                kernel = m.group(1)
            elif BAD.search(line):
                hits.setdefault(kernel, []).append(line.strip())
            elif LNBWD_KERNEL.search(kernel) and AGPR_LDS_STORE.search(line):
                hits.setdefault(kernel + " [AGPR-sourced LDS store in the LayerNorm-backward epilogue]", []).append(line.strip())
        n = sum(len(v) for v in hits.values())
        bad_total += n
        print(f"check_isa: {name}: {n} packed-fp32 op_sel low-lane-selects-high-dword instructions / AGPR-sourced LDS stores "
              f"in the LayerNorm-backward epilogue")
        for k, v in hits.items():
            print(f"  {k}: {len(v)} e.g. {v[0]}")
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
