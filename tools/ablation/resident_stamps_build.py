"""Writes tools/_variants/stamps/sg_conv3x3.hip with cycle stamps in conv3x3_resident_kernel's tile loop (barrier,
commit, barrier, plan + epilogue operands, multiply with the next tile's loads, epilogue); read with tools/conv3x3_stamps.py RESIDENT=1."""
import os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, 'ga-gan_amd/csrc/sg_conv3x3.hip')).read()
src = src.replace("constexpr int kThreads = 256;\nconstexpr int TW = 32, TH = 8;",
                  "__device__ unsigned long long g_stamps[64 * 2 * 8];\nconstexpr int kThreads = 256;\nconstexpr int TW = 32, TH = 8;", 1)
def rep(old, new):
    global src
    assert src.count(old) == 1, old[:50]
    src = src.replace(old, new)
rep("    for (; unit < unit_end; unit += stride) {\n        const int n = p_n, a0 = p_a0, b0 = p_b0;      // (decoded when the tile was planned)\n        __syncthreads();   // everyone is done reading the previous tile (and, the first time, the weights are in place)\n        commit();\n        __syncthreads();\n",
'''    unsigned long long t_ph[6] = {0, 0, 0, 0, 0, 0}, tiles_done = 0;
    auto now = [] { return (unsigned long long)__builtin_amdgcn_s_memtime(); };
    const unsigned long long t_begin = now();
    for (; unit < unit_end; unit += stride) {
        const int n = p_n, a0 = p_a0, b0 = p_b0;
        const unsigned long long t0 = now();
        __syncthreads();   // everyone is done reading the previous tile (and, the first time, the weights are in place)
        const unsigned long long t1 = now();
        commit();
        const unsigned long long t2 = now();
        __syncthreads();
        const unsigned long long t3 = now();
''')
rep("        f32x4_t acc[M_TILES][4];\n#pragma unroll\n        for (int m = 0; m < M_TILES; m++)\n#pragma unroll\n            for (int j = 0; j < 4; j++) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};\n        uint4 fa[2][M_TILES], fb[2][4];",
    "        const unsigned long long t4 = now();\n        f32x4_t acc[M_TILES][4];\n#pragma unroll\n        for (int m = 0; m < M_TILES; m++)\n#pragma unroll\n            for (int j = 0; j < 4; j++) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};\n        uint4 fa[2][M_TILES], fb[2][4];")
rep("        // ---- epilogue -------------------------------------------------------------------------------------------\n        float red[M_TILES];",
    "        const unsigned long long t5 = now();\n        float red[M_TILES];")
rep("        if (g.reduce_out) {\n            // lanes l, l+16, l+32, l+48 share their channels; then the four waves through the input image's LDS",
    "        { const unsigned long long t6 = now(); t_ph[0] += t1 - t0; t_ph[1] += t2 - t1; t_ph[2] += t3 - t2; t_ph[3] += t4 - t3; t_ph[4] += t5 - t4; t_ph[5] += t6 - t5; tiles_done++; }\n        if (g.reduce_out) {\n            // lanes l, l+16, l+32, l+48 share their channels; then the four waves through the input image's LDS")
rep("                g.reduce_out[(unsigned)(unit * g.co + tid)] = r;   // unit = (n * tiles_a + ta) * tiles_b + tb\n            }\n        }\n    }\n}",
'''                g.reduce_out[(unsigned)(unit * g.co + tid)] = r;   // unit = (n * tiles_a + ta) * tiles_b + tb
            }
        }
    }
    if (blockIdx.x < 64 * 4 && (blockIdx.x & 3) == 0 && (tid == 0 || tid == 128)) {
        unsigned long long* o = g_stamps + ((blockIdx.x >> 2) * 2 + (tid >> 7)) * 8;
        for (int k = 0; k < 6; k++) o[k] = t_ph[k];
        o[6] = now() - t_begin; o[7] = tiles_done;
    }
}''')
# ADDR=hits | stream: timing-only experiments on the address pattern of the tile loads (results are wrong):
#   hits    every instruction reads the same 4 KB (L1 hits): the cost of issuing alone
#   stream  every instruction reads 64 consecutive packets (8 whole lines), a different KB per instruction and tile
addr = os.environ.get('ADDR', '')
if addr:
    old = "            voff[q] = ok ? (unsigned)(2 * cg * plane + iy * g.wid + px0) * 2u : kOutside;\n        }\n    };\n    auto fetch_issue"
    new = {'hits': "            voff[q] = ok ? (unsigned)(tid * 16) : (unsigned)(tid * 16);\n        }\n    };\n    auto fetch_issue",
           'stream': "            voff[q] = (unsigned)(((a0 / TH * g.tiles_b + b0 / TW) * STEPS + q) * 4096 + tid * 16) + (ok ? 0u : 0u);\n        }\n    };\n    auto fetch_issue"}[addr]
    rep(old, new)
out = 'tools/_variants/stamps' + ('_' + addr if addr else '')
src = src.rstrip() + '''

extern "C" int sg_debug_conv3x3_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sg::g_stamps), sizeof(sg::g_stamps));
}
'''
os.makedirs(os.path.join(root, out), exist_ok=True)
open(os.path.join(root, out, 'sg_conv3x3.hip'), 'w').write(src)
