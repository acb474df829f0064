"""Builds the round-5 experiment variants of libgga_hip.so from patched COPIES of gga_amd/csrc/dense_conv_ws.hip (the product source
carries no ablation switches): run from the repo root after `make -C gga_amd/csrc`.

    python tools_dev/exp_libs/make_ws_variants.py

  libgga_wsnoprod.so    producers of both producer / consumer kernels do nothing but meet the barriers after the prologue
                        (tools_dev/ab_ws_shape.py under tools_dev/run_with_lib.py)
  libgga_wsstamps.so    dense_conv3x3_ws_kernel with s_memtime stamps around the main loop and the epilogue of every tile +
                        gga_debug_ws_stamps (tools_dev/ws_stamps.py)
"""
import glob
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(REPO, 'gga_amd', 'csrc')
OUT = os.path.join(REPO, 'tools_dev', 'exp_libs')
TMP = '/tmp/gga_ws_variants'
os.makedirs(TMP, exist_ok=True)
base = open(os.path.join(SRC, 'dense_conv_ws.hip')).read()


def build(name, text):
    path = os.path.join(TMP, name + '.hip')
    open(path, 'w').write(text)
    obj = os.path.join(TMP, name + '.o')
    subprocess.check_call(['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wno-unused-function', '-I' + SRC, '-c', '-o', obj, path])
    others = [o for o in glob.glob(os.path.join(SRC, '*.o')) if os.path.basename(o) != 'dense_conv_ws.o']
    subprocess.check_call(['hipcc', '-shared', '-fPIC', '--offload-arch=gfx950', '-o', os.path.join(OUT, f'libgga_{name}.so')] + others + [obj])
    print('built', name)


# ---- producers idle
s = base
a = s.index('#define WS_STAGE_P(TAP, CH, HB, INCUR2, L0, L1, S0, S1) {')
b = s.index('#define WS_EVEN(TAP, CH, HB, INCUR2) WS_STAGE_P')
s = s[:a] + '#define WS_STAGE_P(TAP, CH, HB, INCUR2, L0, L1, S0, S1) { __syncthreads(); }\n' + s[b:]
a = s.index('                    WS_BST(2 * ((k + 1) % 3) + 0, wq[(k + 1) & 1][0][0], wq[(k + 1) & 1][0][1])')
b = s.index('                    __syncthreads();\n                }\n            }\n#pragma unroll\n            for (int e = 0; e < NA; ++e) pcur[e] = pnxt[e];', a)
build('wsnoprod', s[:a] + s[b:])

# ---- stamps in the 32x32x16 kernel
s = base
a = s.index('template <int NT, int MT>\n__global__ __launch_bounds__(512, 2) void dense_conv3x3_ws_kernel')
s = s[:a] + ('__device__ unsigned long long ws_stamp[256 * 4];\n'
             'extern "C" int gga_debug_ws_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ws_stamp), sizeof(unsigned long long) * 1024); }\n') + s[a:]
b = s.index('    for (; tile < n_tiles; tile += gridDim.x) {\n        const int slice = tile / img_tiles, itile = tile - slice * img_tiles;')
s = s[:b] + '    unsigned long long st_main = 0, st_epi = 0, st_tiles = 0; const unsigned long long st_begin = __builtin_amdgcn_s_memtime();\n' + s[b:]
c = s.index('        WS_READ_A(fa, 0, 0)\n        WS_READ_B(fb, 0)\n#define WS_STAGE_C')
s = s[:c] + '        const unsigned long long st0 = __builtin_amdgcn_s_memtime();\n' + s[c:]
d = s.index('        // ---- epilogue (consumers): back from the scaled operands')
s = s[:d] + '        const unsigned long long st1 = __builtin_amdgcn_s_memtime();\n' + s[d:]
e = s.index('    }\n    if (stats) { WS_FLUSH_STATS() }\n#undef WS_FLUSH_STATS\n#undef WS_READ_A')
s = s[:e] + '        { const unsigned long long st2 = __builtin_amdgcn_s_memtime(); st_main += st1 - st0; st_epi += st2 - st1; st_tiles += 1; }\n' + s[e:]
f = s.index('    if (stats) { WS_FLUSH_STATS() }\n#undef WS_FLUSH_STATS\n#undef WS_READ_A')
s = s[:f] + ('    if (tid == 0) { ws_stamp[blockIdx.x * 4 + 0] = st_main; ws_stamp[blockIdx.x * 4 + 1] = st_epi; ws_stamp[blockIdx.x * 4 + 2] = st_tiles; '
             'ws_stamp[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime() - st_begin; }\n') + s[f:]
build('wsstamps', s)
stamped = s

# ---- stamps + producers idle (+ no consumer stage barrier / fragments read once): where the main loop's cycles go
def noprod(t):
    a = t.index('#define WS_STAGE_P(TAP, CH, HB, INCUR2, L0, L1, S0, S1) {')
    b = t.index('#define WS_EVEN(TAP, CH, HB, INCUR2) WS_STAGE_P')
    return t[:a] + '#define WS_STAGE_P(TAP, CH, HB, INCUR2, L0, L1, S0, S1) { __syncthreads(); }\n' + t[b:]


build('wsstamps_noprod', noprod(stamped))
# every halo piece from the zero page (an L2-resident kilobyte): the producers' vector-memory queue without HBM latency in it
build('wsstamps_halozero', stamped.replace('P[e] = ok ? xb_ + ((iy * prow + ix * pcol) * cin + q * 4) : zero_page;', 'P[e] = zero_page;'))
# weights from stage 0 always (L2-hot 8 KB) - the other half of the producers' loads
build('wsstamps_w0', stamped.replace('const uint4* bsrc = reinterpret_cast<const uint4*>((WP) + ((int64_t)(TAP) * nchunks + (CH)) * (2 * CO * DC_CK));', 'const uint4* bsrc = reinterpret_cast<const uint4*>((WP) + ((int64_t)(TAP) * 0 + 0 * (CH)) * (2 * CO * DC_CK));'))
t = noprod(stamped).replace('            WS_CONSUMER_BARRIER(cons_a, cons_b) }', '            }')
t = t.replace('#define WS_STAGE_P(TAP, CH, HB, INCUR2, L0, L1, S0, S1) { __syncthreads(); }', '#define WS_STAGE_P(TAP, CH, HB, INCUR2, L0, L1, S0, S1) { }')
build('wsstamps_noprod_nobar', t)
t2 = t.replace('            WS_READ_A(XA, ((TAP) + 1) % 9, (TAP) == 8 ? 1 - (HB) : (HB))                                              \\\n            WS_READ_B(XB, ((TAP) + 1) % 3)                                                                            \\\n', '            XA[0][0] = CA[0][0]; XB[0][0] = CB[0][0];                                                             \\\n')
build('wsstamps_noprod_nobar_noread', t2)
