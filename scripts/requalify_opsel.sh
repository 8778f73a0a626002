#!/bin/bash
# Re-qualification of the packed-fp32 op_sel finding after a toolchain bump -- ONE command:
#   scripts/requalify_opsel.sh            (on a GPU box; builds what is missing with the box's hipcc)
# 1. scripts/probe/pk_opsel_repro.hip: the two instruction forms alone, one launch of a large grid, bad / pinned / scalar
#    compared lane by lane (consumer 0..4 slots behind; plain VALU stream, behind ds_bpermute, behind an MFMA).
# 2. the library's OWN epilogue built without the pin (-DXF_LN_DIAG=32 -DXF_LN_EPI_MIN_WAVES=1: the build that failed
#    12 / 12 in round 2; check_isa.py reports how many such instructions it contains) against the product build, 16 launches
#    each on identical inputs (scripts/probe/lnbwd_determinism.py).
# Prints hipcc / ROCm / device beside the counts. Expected while the finding stands: (1) may pass -- the bare forms did not
# reproduce on their own (round 3: 0 of 1.6e10 evaluations) --, (2) unpinned build differs, product build does not.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p build gpurun_out
echo "== toolchain: $(/opt/rocm/bin/hipcc --version | grep -i -m1 'HIP version') | $(cat /opt/rocm/.info/version 2>/dev/null)"
[ -x build/pk_opsel_repro ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value scripts/probe/pk_opsel_repro.hip -o build/pk_opsel_repro || exit 2
[ -f build/libxfmr_hip_unpinned.so ] || ISA_AUDIT=0 scripts/build_variant.sh unpinned "-DXF_LN_DIAG=32 -DXF_LN_EPI_MIN_WAVES=1" gemm.o || exit 2
grep -h "check_isa: gemm.o" /tmp/xfbuild_unpinned/make.log 2>/dev/null
echo "== 1. the instruction forms alone"
build/pk_opsel_repro; rc1=$?
echo "== 2a. the epilogue WITHOUT the pin (expected: launches differ)"
XFMR_HIP_LIB=$ROOT/build/libxfmr_hip_unpinned.so timeout -k 10 300 python3 scripts/probe/lnbwd_determinism.py 2>&1 | grep -E "launches differ|library|rows affected" 
echo "== 2b. the product build (expected: 0 of 16)"
timeout -k 10 300 python3 scripts/probe/lnbwd_determinism.py 2>&1 | grep -E "launches differ|library"
echo "standalone probe exit status: $rc1"
