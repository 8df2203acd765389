"""Writes tools/_variants/fstamps/sg_conv2d.hip with cycle stamps in the stage loop of conv2d_mfma_kernel's float path
(issue of the next stage's loads, multiply, footprint commit, wait for the copies, barrier); read with
tools/float_stamps.py."""
import os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, 'ga-gan_amd/csrc/sg_conv2d.hip')).read()
def rep(old, new):
    global src
    assert src.count(old) == 1, old[:60]
    src = src.replace(old, new)
rep("constexpr int kPixels = 256;      // output pixels per workgroup",
    "__device__ unsigned long long g_fstamps[64 * 2 * 8];\nconstexpr int kPixels = 256;      // output pixels per workgroup")
rep('''        int sidx = 0;                       // stage counter: picks the weight buffer
        if (cb < cb_end) {''', '''        int sidx = 0;                       // stage counter: picks the weight buffer
        unsigned long long t_ph[6] = {0, 0, 0, 0, 0, 0}, stages = 0;
        auto now = [] { return (unsigned long long)__builtin_amdgcn_s_memtime(); };
        const unsigned long long t_begin = now();
        if (cb < cb_end) {''')
rep('''            const bool first = g0 == 0, last = cb1 != cb;
            if (cb1 < cb_end) w_issue(cb1, g1, s_wb(sidx + 1));
            if (first && cb + 1 < cb_end) x_load(cb + 1);
            split_multiply(s_wb(sidx), s_x(cb), g0, min(kSplitTaps, taps - g0));
            if (last && cb1 < cb_end) {
                if (!t.x_double) __syncthreads();   // every wave has read the footprint of this chunk
                x_commit(s_x(cb1), cb1);
            }
            copies_landed();
            __syncthreads();
''', '''            const bool first = g0 == 0, last = cb1 != cb;
            const unsigned long long t0 = now();
            if (cb1 < cb_end) w_issue(cb1, g1, s_wb(sidx + 1));
            if (first && cb + 1 < cb_end) x_load(cb + 1);
            const unsigned long long t1 = now();
            split_multiply(s_wb(sidx), s_x(cb), g0, min(kSplitTaps, taps - g0));
            const unsigned long long t2 = now();
            if (last && cb1 < cb_end) {
                if (!t.x_double) __syncthreads();   // every wave has read the footprint of this chunk
                x_commit(s_x(cb1), cb1);
            }
            const unsigned long long t3 = now();
            copies_landed();
            const unsigned long long t4 = now();
            __syncthreads();
            const unsigned long long t5 = now();
            t_ph[0] += t1 - t0; t_ph[1] += t2 - t1; t_ph[2] += t3 - t2; t_ph[3] += t4 - t3; t_ph[4] += t5 - t4; stages++;
''')
rep('''            advance(cb1, g1);
            sidx++;
        }
    }
''', '''            advance(cb1, g1);
            sidx++;
        }
        const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (wg < 64 * 4 && (wg & 3) == 0 && (tid == 0 || tid == 256)) {
            unsigned long long* o = g_fstamps + ((wg >> 2) * 2 + (tid >> 8)) * 8;
            for (int k = 0; k < 5; k++) o[k] = t_ph[k];
            o[5] = now() - t_begin; o[6] = stages; o[7] = t_begin;
        }
    }
''')
src = src.rstrip() + '''

extern "C" int sg_debug_conv2d_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sg::g_fstamps), sizeof(sg::g_fstamps));
}
'''
os.makedirs(os.path.join(root, 'tools/_variants/fstamps'), exist_ok=True)
open(os.path.join(root, 'tools/_variants/fstamps/sg_conv2d.hip'), 'w').write(src)
