"""Writes stubs.cpp: a no-op definition of every kernel launcher declared in csrc/kernels.hpp (the host-only sanitizer
build links the C-ABI's host code against these instead of the HIP kernels).  Launchers whose RESULT the host code reads
back get a minimal host implementation: launch_text_lens_scan (dense rows)."""
import re
import sys
from pathlib import Path

src = Path(sys.argv[1]).read_text()
src = re.sub(r"//[^\n]*", "", src)
out = ['#include "kernels.hpp"', ""]
GEMM_CHECK = r'''{
    // the operand / output ranges the kernels would touch must lie inside live device blocks (catches arena, plane-offset and
    // padded-row arithmetic slips of the callers)
    auto bad = [&](const char* what) { fprintf(stderr, "hip stub: launch_gemm_bf16 %s range outside its device block (I=%d J=%d K=%d planes=%d)\n", what, L.I, L.J, L.K, L.planes); abort(); };
    if (L.I <= 0 || L.J <= 0 || L.K <= 0 || L.K % 64 || L.planes < 1 || L.planes > 9) return hipErrorInvalidValue;
    const int64_t ra = L.a_rows_padded ? ((int64_t)L.I + 255) / 256 * 256 : L.I, rb = L.b_rows_padded ? ((int64_t)L.J + 255) / 256 * 256 : L.J;
    const int64_t lda = L.lda ? L.lda : L.K, ldb = L.ldb ? L.ldb : L.K;
    for (int p = 0; p < L.planes; ++p) {
        if (!hip_stub_range_ok(L.A + L.a_plane_off[p], (size_t)((ra - 1) * lda + L.K) * 2, true)) bad("A");
        if (!hip_stub_range_ok(L.B + L.b_plane_off[p], (size_t)((rb - 1) * ldb + L.K) * 2, true)) bad("B");
    }
    const size_t es = (L.epilogue == TVC_EPI_F32 || L.epilogue == TVC_EPI_RESID_F32) ? 4 : 2;
    if (!hip_stub_range_ok(L.out, (size_t)(((int64_t)L.J - 1) * L.ldo + L.I) * es, true)) bad("out");
    if (L.bias && !hip_stub_range_ok(L.bias, (size_t)L.I * 4, true)) bad("bias");
    if (L.splitk_fixed >= 2) {
        const size_t tiles = (size_t)((L.I + 255) / 256) * ((L.J + 255) / 256);
        if (!L.splitk_ws || tiles * L.splitk_fixed * 256 * 256 * 4 > L.splitk_ws_bytes || !hip_stub_range_ok(L.splitk_ws, L.splitk_ws_bytes, true)) bad("split-K workspace");
    }
    return hipSuccess;
}'''
special = {
    "launch_gemm_bf16": GEMM_CHECK,
    "launch_text_lens_scan": "{ for (int i = 0; i <= n_text; ++i) starts[i] = i * ctx; starts[n_text + 1] = ctx;"
                             " if (pfx) for (int i = 0; i < n_text; ++i) { pfx[i] = 0; pfx[n_text + i] = i * ctx; } return hipSuccess; }",
}
for m in re.finditer(r"\bhipError_t\s+(\w+)\s*\(([^;{]*)\)\s*;", src):
    name, args = m.group(1), " ".join(m.group(2).split())
    args = re.sub(r"\s*=\s*[^,)]+", "", args)                # default arguments belong to the declaration
    body = special.get(name, "{ return hipSuccess; }")
    out.append(f"hipError_t {name}({args}) {body}")
Path(sys.argv[2]).write_text("\n".join(out) + "\n")
print(len(out) - 2, "stubs")
