"""Writes tools/_variants/stamps/sg_conv3x3.hip: the product kernel with cycle stamps around the phases of its K loop
(barrier, commit, barrier, multiply) and a host entry point that returns them; tools/conv3x3_stamps.py reads them.
Build: hipcc -c that file, link it with the other objects of ga-gan_amd/build into tools/_variants/lib_stamps.so."""
import os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, 'ga-gan_amd/csrc/sg_conv3x3.hip')).read()
src = src.replace("constexpr int kThreads = 256;\nconstexpr int TW = 32, TH = 8;",
                  "__device__ unsigned long long g_stamps[64 * 2 * 8];\nconstexpr int kThreads = 256;\nconstexpr int TW = 32, TH = 8;", 1)
old = '''        __syncthreads();   // everyone is done reading the previous chunk
        commit();
        __syncthreads();
        fetch_begin(c0 + KCH);
        multiply(std::true_type{});      // (the next chunk's loads are issued between its taps)
    }
'''
new = '''        const unsigned long long t0 = now();
        __syncthreads();   // everyone is done reading the previous chunk
        const unsigned long long t1 = now();
        commit();
        const unsigned long long t2 = now();
        __syncthreads();
        const unsigned long long t3 = now();
        fetch_begin(c0 + KCH);
        const unsigned long long t4 = now();
        multiply(std::true_type{});
        const unsigned long long t5 = now();
        t_bar1 += t1 - t0; t_commit += t2 - t1; t_bar2 += t3 - t2; t_fetch += t4 - t3; t_mul += t5 - t4;
    }
'''
assert src.count(old) == 1
src = src.replace(old, new)
old = "    fetch(0);\n    const int c_last = ((g.ci - 1) / KCH) * KCH;\n"
new = '''    unsigned long long t_bar1 = 0, t_commit = 0, t_bar2 = 0, t_fetch = 0, t_mul = 0;
    auto now = [] { return (unsigned long long)__builtin_amdgcn_s_memtime(); };
    const unsigned long long t_begin = now(), r_begin = __builtin_amdgcn_s_memrealtime();
    fetch(0);
    const int c_last = ((g.ci - 1) / KCH) * KCH;
'''
assert src.count(old) == 1
src = src.replace(old, new)
old = "    multiply(std::false_type{});\n"
new = old + '''    if (blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && (tid == 0 || tid == 256)) {
        unsigned long long* o = g_stamps + ((blockIdx.x >> 3) * 2 + (tid >> 8)) * 8;
        o[0] = t_bar1; o[1] = t_commit; o[2] = t_bar2; o[3] = t_fetch; o[4] = t_mul;
        o[5] = now() - t_begin; o[6] = __builtin_amdgcn_s_memrealtime() - r_begin; o[7] = (unsigned long long)(c_last / KCH);
    }
'''
assert src.count(old) == 1
src = src.replace(old, new)
src = src.rstrip() + '''

extern "C" int sg_debug_conv3x3_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sg::g_stamps), sizeof(sg::g_stamps));
}
'''
os.makedirs(os.path.join(root, 'tools/_variants/stamps'), exist_ok=True)
open(os.path.join(root, 'tools/_variants/stamps/sg_conv3x3.hip'), 'w').write(src)
