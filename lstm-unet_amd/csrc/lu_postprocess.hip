// lu_postprocess.hip -- inference post-processing on the GPU: softmax [3,H,W] -> instance label map, the device side of
// Inference2D.postprocess (reference Inference2D.py:66-123, host numpy / scipy / OpenCV there).
//
// Integer / index work, bit-exact by construction; everything is HBM- or latency-bound (one 832 x 992 frame is 0.8 M pixels),
// nothing here wants the matrix pipe.  Building blocks:
//   classify            edge = softmax[2] >= 0.2, cell = (argmax == 1) & !edge                                   (:66-68)
//   union-find CCL      one kernel family for all four labelling problems (connectivity and the "same set" predicate are
//                       mode parameters): background 4-conn (binary_fill_holes :69), foreground 8-conn
//                       (cv2.connectedComponentsWithStats :72), equal-label 8-conn (component count per label), and
//                       "everything but label n" 4-conn inside a crop (the per-object binary_fill_holes :86).
//                       parent[p] = p; every pixel merges with its already-scanned neighbours through atomicMin on the
//                       roots (the smaller linear index wins, so a component's root is its first pixel in raster order and
//                       the result does not depend on scheduling); a last pass flattens the trees.
//   OpenCV label order  components are numbered by the block-raster position of their first 2 x 2 block (why: see
//                       oracle/postprocess_oracle.py): atomicMin of the block index per root, a 0/1 mark per block, an
//                       exclusive scan over the block grid = label - 1.  Areas by atomicAdd (integers: order-free).
//   edge absorption     an edge pixel closer than edge_dist to a cell pixel takes the label of the NEAREST cell pixel with
//                       scipy's distance_transform_edt tie-break (smallest column, then smallest row: pinned against scipy
//                       in the tests) -- a window search, embarrassingly parallel                                (:77-78)
//   per-label statistics bounding boxes, component counts and the bit-quad Euler number E8 = (Q1 - Q3 - 2 QD) / 4 of every
//                       label at once: holes(n) = components(n) - E8(n) tells the host which objects have holes at all
//   object hole fill    CCL of "not n" inside the object's bounding box (+1 pixel), components that do not reach the box
//                       border are the holes; L += n there -- the reference's additive quirk included; a `dirty` flag
//                       reports a hole that contained another label (the host then falls back to the strictly sequential
//                       order of the reference for the remaining labels)                                          (:80-91)
//   device-driven tail  fill_all_kernel fills the holes of ALL objects in one launch (one workgroup per object, crop flooded
//                       in LDS) and newid_kernel numbers the kept labels, so that a frame needs no host decision and ONE
//                       device -> host copy; the per-object path above remains the exact fallback for nested objects
//   FOV presence / relabel   which labels survive the field-of-view mask (single-column quirk of :97 selectable), final
//                       consecutive ids as uint16                                                              (:93-123)
#include <stdint.h>
#include <string.h>
#include "lu_device.h"

namespace {

constexpr int PT = 256;
enum { MODE_BG4 = 0, MODE_FG8 = 1, MODE_EQ8 = 2, MODE_NE4 = 3 };

inline unsigned pgrid(int64_t n) {
    int64_t b = (n + PT - 1) / PT;
    if (b < 1) b = 1;
    if (b > 65535) b = 65535;
    return (unsigned)b;
}

struct Rect {
    int x0, y0, w, h;      // crop inside the H x W image (row stride W)
};

__global__ void classify_kernel(const float* __restrict__ sm, int64_t hw, float thresh, int32_t* __restrict__ cell,
                                int32_t* __restrict__ edge) {
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < hw; i += (int64_t)gridDim.x * PT) {
        const float a = sm[i], b = sm[hw + i], c = sm[2 * hw + i];
        const int e = c >= thresh;
        edge[i] = e;
        cell[i] = (b > a && b >= c && !e) ? 1 : 0;      // np.argmax: the FIRST maximum wins
    }
}

__device__ __forceinline__ bool ccl_active(int v, int mode, int n, int nl) {
    switch (mode) {
        case MODE_BG4: return v == 0;
        case MODE_FG8: return v != 0;
        case MODE_EQ8: return v > 0 && v < nl;
        default: return v != n;
    }
}

__device__ __forceinline__ int ccl_find(const int32_t* parent, int i) {
    int p = parent[i];
    while (p != i) {
        i = p;
        p = parent[i];
    }
    return i;
}

__device__ __forceinline__ void ccl_union(int32_t* parent, int a, int b) {
    while (true) {
        a = ccl_find(parent, a);
        b = ccl_find(parent, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&parent[a], b);      // a was a root: hang it under the smaller root
        if (old == a) return;
        a = old;                                       // somebody re-parented a meanwhile: continue from there
    }
}

__global__ void ccl_init_kernel(const int32_t* __restrict__ val, int32_t* __restrict__ parent, int W, Rect r, int mode, int n,
                                int nl) {
    const int64_t total = (int64_t)r.w * r.h;
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < total; i += (int64_t)gridDim.x * PT) {
        const int y = r.y0 + (int)(i / r.w), x = r.x0 + (int)(i % r.w);
        const int p = y * W + x;
        parent[p] = ccl_active(val[p], mode, n, nl) ? p : -1;
    }
}

__global__ void ccl_merge_kernel(const int32_t* __restrict__ val, int32_t* parent, int W, Rect r, int mode, int n, int nl) {
    const int64_t total = (int64_t)r.w * r.h;
    const bool eight = mode == MODE_FG8 || mode == MODE_EQ8;
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < total; i += (int64_t)gridDim.x * PT) {
        const int ly = (int)(i / r.w), lx = (int)(i % r.w);
        const int p = (r.y0 + ly) * W + r.x0 + lx;
        const int v = val[p];
        if (!ccl_active(v, mode, n, nl)) continue;
        // already-scanned neighbours: left, up (+ up-left, up-right for 8-connectivity), inside the crop
        const int dxs[4] = {-1, 0, -1, 1}, dys[4] = {0, -1, -1, -1};
        for (int k = 0; k < (eight ? 4 : 2); ++k) {
            const int qx = lx + dxs[k], qy = ly + dys[k];
            if (qx < 0 || qx >= r.w || qy < 0) continue;
            const int q = (r.y0 + qy) * W + r.x0 + qx;
            const int u = val[q];
            if (!ccl_active(u, mode, n, nl) || (mode == MODE_EQ8 && u != v)) continue;
            ccl_union(parent, p, q);
        }
    }
}

__global__ void ccl_flatten_kernel(int32_t* parent, int W, Rect r) {
    const int64_t total = (int64_t)r.w * r.h;
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < total; i += (int64_t)gridDim.x * PT) {
        const int p = (r.y0 + (int)(i / r.w)) * W + r.x0 + (int)(i % r.w);
        if (parent[p] >= 0) parent[p] = ccl_find(parent, p);
    }
}

// flag[root] = 1 for components touching the border of the crop
__global__ void border_flag_kernel(const int32_t* __restrict__ parent, int32_t* __restrict__ flag, int W, Rect r) {
    const int64_t total = 2 * (int64_t)(r.w + r.h);
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < total; i += (int64_t)gridDim.x * PT) {
        int lx, ly;
        if (i < r.w) { lx = (int)i; ly = 0; }
        else if (i < 2 * r.w) { lx = (int)(i - r.w); ly = r.h - 1; }
        else if (i < 2 * r.w + r.h) { lx = 0; ly = (int)(i - 2 * r.w); }
        else { lx = r.w - 1; ly = (int)(i - 2 * r.w - r.h); }
        const int p = (r.y0 + ly) * W + r.x0 + lx;
        const int root = parent[p];
        if (root >= 0) flag[root] = 1;
    }
}

__global__ void fill_bg_kernel(int32_t* __restrict__ cell, const int32_t* __restrict__ parent, const int32_t* __restrict__ flag,
                               int32_t* __restrict__ edge, int64_t hw) {
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < hw; i += (int64_t)gridDim.x * PT) {
        const int root = parent[i];
        if (root >= 0 && !flag[root]) cell[i] = 1;      // a background component that never reaches the frame border
        if (cell[i]) edge[i] = 0;                       // seg_edge = max(seg_edge - seg_cell, 0)
    }
}

__global__ void bbox_init_kernel(int32_t* __restrict__ bbox, int64_t n) {      // min fields INT_MAX, max fields -1
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < n; i += (int64_t)gridDim.x * PT) {
        bbox[4 * i] = bbox[4 * i + 1] = 0x7fffffff;
        bbox[4 * i + 2] = bbox[4 * i + 3] = -1;
    }
}

__global__ void fill32_kernel(int32_t* __restrict__ a, int32_t v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < n; i += (int64_t)gridDim.x * PT) a[i] = v;
}

__global__ void copy32_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < n; i += (int64_t)gridDim.x * PT) dst[i] = src[i];
}

__global__ void fg_key_kernel(const int32_t* __restrict__ parent, int32_t* __restrict__ key, int H, int W) {
    const int64_t hw = (int64_t)H * W;
    const int bw = (W + 1) / 2;
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < hw; i += (int64_t)gridDim.x * PT) {
        const int root = parent[i];
        if (root < 0) continue;
        const int y = (int)(i / W), x = (int)(i % W);
        atomicMin(&key[root], (y >> 1) * bw + (x >> 1));
    }
}

__global__ void key_mark_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ key, int32_t* __restrict__ mark,
                                int64_t hw) {
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < hw; i += (int64_t)gridDim.x * PT)
        if (parent[i] == i) mark[key[i]] = 1;
}

// exclusive prefix sum of `in` (n entries) by ONE block of 1024 threads; total -> *total_out
__global__ __launch_bounds__(1024) void scan_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int n,
                                                    int32_t* total_out) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = lo + per < n ? lo + per : n;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += in[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t ? part[t - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        const int v = in[i];
        out[i] = run;
        run += v;
    }
    if (t == 1023) *total_out = part[1023] + 1;      // number of labels INCLUDING the background
}

__global__ void fg_assign_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ key,
                                 const int32_t* __restrict__ scanv, int32_t* __restrict__ lab, int32_t* __restrict__ area,
                                 int64_t hw) {
    // area[0] (the background: most of the frame) is counted per block in LDS and added once -- one global atomic per PIXEL
    // on that single address serialised the whole kernel (0.72 ms of a 1.2 ms post-processing chain at 256 x 256)
    __shared__ int32_t bg;
    if (threadIdx.x == 0) bg = 0;
    __syncthreads();
    // a thread walks 8 consecutive pixels and adds RUNS of equal labels (the pixels of an object are row-contiguous)
    const int64_t runs = (hw + 7) / 8;
    for (int64_t t = (int64_t)blockIdx.x * PT + threadIdx.x; t < runs; t += (int64_t)gridDim.x * PT) {
        int cur = -1, cnt = 0;
        const int64_t hi = (t + 1) * 8 < hw ? (t + 1) * 8 : hw;
        for (int64_t i = t * 8; i < hi; ++i) {
            const int root = parent[i];
            const int l = root >= 0 ? scanv[key[root]] + 1 : 0;
            lab[i] = l;
            if (l != cur) {
                if (cnt) {
                    if (cur) atomicAdd(&area[cur], cnt);
                    else atomicAdd(&bg, cnt);
                }
                cur = l;
                cnt = 0;
            }
            ++cnt;
        }
        if (cnt) {
            if (cur) atomicAdd(&area[cur], cnt);
            else atomicAdd(&bg, cnt);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && bg) atomicAdd(&area[0], bg);
}

__global__ void absorb_kernel(const int32_t* __restrict__ cc, const int32_t* __restrict__ edge, int32_t* __restrict__ lab,
                              int H, int W, double edge_dist, int radius) {
    const int64_t hw = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < hw; i += (int64_t)gridDim.x * PT) {
        int l = cc[i];
        if (l == 0 && edge[i]) {
            const int y = (int)(i / W), x = (int)(i % W);
            int bd = 0x7fffffff, bx = 0, by = 0, bl = 0;
            for (int dy = -radius; dy <= radius; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -radius; dx <= radius; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    const int v = cc[(int64_t)yy * W + xx];
                    if (!v) continue;
                    const int d2 = dy * dy + dx * dx;
                    // nearest; ties: smallest column, then smallest row (scipy's feature transform)
                    if (d2 < bd || (d2 == bd && (xx < bx || (xx == bx && yy < by)))) {
                        bd = d2; bx = xx; by = yy; bl = v;
                    }
                }
            }
            if (bl && sqrt((double)bd) < edge_dist) l = bl;
        }
        lab[i] = l;
    }
}

// per-label bounding boxes and 4 * Euler number (bit quads over the zero-padded frame); labels outside [1, nl) are ignored
__global__ void label_stats_kernel(const int32_t* __restrict__ lab, int H, int W, int nl, int32_t* __restrict__ bbox,
                                   int32_t* __restrict__ e4) {
    const int64_t total = (int64_t)(H + 1) * (W + 1);
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < total; i += (int64_t)gridDim.x * PT) {
        const int y = (int)(i / (W + 1)), x = (int)(i % (W + 1));      // quad = pixels (y-1..y, x-1..x)
        int q[4];
        q[0] = (y > 0 && x > 0) ? lab[(int64_t)(y - 1) * W + x - 1] : 0;
        q[1] = (y > 0 && x < W) ? lab[(int64_t)(y - 1) * W + x] : 0;
        q[2] = (y < H && x > 0) ? lab[(int64_t)y * W + x - 1] : 0;
        q[3] = (y < H && x < W) ? lab[(int64_t)y * W + x] : 0;
        if (y < H && x < W) {
            const int v = q[3];
            if (v > 0 && v < nl) {      // only a pixel whose neighbour on that side is NOT v can be the extreme on that side
                if (q[2] != v) atomicMin(&bbox[4 * v], x);
                if (q[1] != v) atomicMin(&bbox[4 * v + 1], y);
                if (x + 1 >= W || lab[(int64_t)y * W + x + 1] != v) atomicMax(&bbox[4 * v + 2], x);
                if (y + 1 >= H || lab[(int64_t)(y + 1) * W + x] != v) atomicMax(&bbox[4 * v + 3], y);
            }
        }
        for (int a = 0; a < 4; ++a) {
            const int v = q[a];
            if (v <= 0 || v >= nl) continue;
            bool seen = false;
            for (int b = 0; b < a; ++b) seen = seen || q[b] == v;
            if (seen) continue;
            const int m = (q[0] == v) | ((q[1] == v) << 1) | ((q[2] == v) << 2) | ((q[3] == v) << 3);
            const int cnt = __popc(m);
            int d = 0;
            if (cnt == 1) d = 1;
            else if (cnt == 3) d = -1;
            else if (m == 0x9 || m == 0x6) d = -2;      // the two diagonal quads
            if (d) atomicAdd(&e4[v], d);
        }
    }
}

__global__ void count_roots_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ lab, int32_t* __restrict__ ncomp,
                                   int64_t hw) {
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < hw; i += (int64_t)gridDim.x * PT)
        if (parent[i] == i) atomicAdd(&ncomp[lab[i]], 1);
}

__global__ void clear_flag_kernel(const int32_t* __restrict__ parent, int32_t* __restrict__ flag, int W, Rect r) {
    const int64_t total = (int64_t)r.w * r.h;
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < total; i += (int64_t)gridDim.x * PT) {
        const int p = (r.y0 + (int)(i / r.w)) * W + r.x0 + (int)(i % r.w);
        flag[p] = 0;
    }
}

// holes of object n inside the crop: "not n" components without a border flag get += n
__global__ void object_fill_kernel(int32_t* lab, const int32_t* __restrict__ parent, const int32_t* __restrict__ flag, int W,
                                   Rect r, int n, int32_t* dirty) {
    const int64_t total = (int64_t)r.w * r.h;
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < total; i += (int64_t)gridDim.x * PT) {
        const int p = (r.y0 + (int)(i / r.w)) * W + r.x0 + (int)(i % r.w);
        const int root = parent[p];
        if (root >= 0 && !flag[root]) {
            const int v = lab[p];
            if (v != 0) *dirty = 1;      // the hole held another label: later objects may see changed sets
            lab[p] = v + n;
        }
    }
}

__global__ void bbox_of_label_kernel(const int32_t* __restrict__ lab, int H, int W, int n, int32_t* __restrict__ box) {
    const int64_t hw = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < hw; i += (int64_t)gridDim.x * PT) {
        if (lab[i] != n) continue;
        const int y = (int)(i / W), x = (int)(i % W);
        atomicMin(&box[0], x);
        atomicMin(&box[1], y);
        atomicMax(&box[2], x);
        atomicMax(&box[3], y);
    }
}

__global__ void present_kernel(const int32_t* __restrict__ lab, int H, int W, int fov, int single_column, int nl,
                               int32_t* __restrict__ present) {
    const int64_t hw = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < hw; i += (int64_t)gridDim.x * PT) {
        const int y = (int)(i / W), x = (int)(i % W);
        // fov_im[:fov, :] = 0; fov_im[-fov:, :] = 0; fov_im[:, fov] = 0 (reference quirk) or [:, :fov]; fov_im[:, -fov:] = 0
        const bool out = y < fov || y >= H - fov || (single_column ? x == fov : x < fov) || x >= W - fov;
        const int v = out ? 0 : lab[i];
        if (v >= 0 && v < nl) present[v] = 1;
    }
}

__global__ void relabel_kernel(const int32_t* __restrict__ lab, const int32_t* __restrict__ newid, int nl,
                               unsigned short* __restrict__ out, int64_t hw) {
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < hw; i += (int64_t)gridDim.x * PT) {
        const int v = lab[i];
        out[i] = (v > 0 && v < nl) ? (unsigned short)newid[v] : (unsigned short)0;
    }
}

// ---- device-driven tail of the frame (no host decision between label statistics and the final map) --------------------
// Hole filling of ALL objects in one launch: a workgroup takes the labels v with holes (components - Euler number > 0, read
// from the device arrays) one at a time, stages the object's crop (bounding box + 1 pixel, clipped) in LDS as one byte per
// pixel -- 0 wall (== v), 1 open, 2 open and connected (4-conn) to the crop border -- and floods from the border with row /
// column sweeps until nothing changes; open pixels the flood never reached are the holes: L += v.
// Objects are processed concurrently although the reference walks them in label order: when no hole holds a non-zero label
// all holes are disjoint regions of zeros, so the order cannot matter.  When one does (nested objects: the additive quirk), or a
// crop does not fit the LDS staging, flags[0] is set: this result is discarded and the reference's strict order replayed from
// the snapshot of the map by fill_sequential_kernel (one workgroup, still on the device).  A pixel another workgroup fills concurrently is read as 0 or as that label -- "not v" either way.
constexpr int FILL_MAX_PIX = 48 * 1024;

// One object: stage the crop of label v (bbox + 1 pixel, clipped) in LDS, flood the "not v" pixels from the crop border, add v
// to what the flood never reached.  -> 0 ok, 1 crop too large (nothing done).  *hit is set when a hole held a non-zero label.
// GROW (the sequential replay): a pixel that changes from m to m + v extends label (m + v)'s bounding box in the table, so that
// later labels are cropped from boxes that CONTAIN their current pixel set (a box that is too large is harmless: everything
// outside an object's true box is "not v" and connected to the crop border, so the same components are holes).
// A crop beyond the LDS staging (FILL_MAX_PIX) is flooded in `big` (global scratch of `big_cap` bytes, sequential replay only).
template <bool GROW>
__device__ __forceinline__ int fill_one_object(int32_t* lab, int H, int W, int v, int32_t* bbox, int nmax, unsigned char* st,
                                               int* changed, int* hit, unsigned char* big = nullptr, int64_t big_cap = 0,
                                               int32_t* dirty = nullptr) {
    const int tid = threadIdx.x;
    const int bx0 = bbox[4 * v], by0 = bbox[4 * v + 1], bx1 = bbox[4 * v + 2], by1 = bbox[4 * v + 3];
    if (bx1 < bx0 || by1 < by0) return 0;      // the label does not occur (`if not np.any(bw): continue`)
    const int x0 = bx0 > 0 ? bx0 - 1 : 0, y0 = by0 > 0 ? by0 - 1 : 0;
    const int x1 = bx1 < W - 1 ? bx1 + 1 : W - 1, y1 = by1 < H - 1 ? by1 + 1 : H - 1;
    const int w = x1 - x0 + 1, h = y1 - y0 + 1;
    if ((int64_t)w * h > FILL_MAX_PIX) {
        if (!big || (int64_t)w * h > big_cap) return 1;
        st = big;
    }
    for (int i = tid; i < w * h; i += PT) {
        const int y = i / w, x = i - y * w;
        const int val = lab[(int64_t)(y0 + y) * W + x0 + x];
        st[i] = val == v ? 0 : ((x == 0 || y == 0 || x == w - 1 || y == h - 1) ? 2 : 1);
    }
    __syncthreads();
    for (;;) {
        if (tid == 0) *changed = 0;
        __syncthreads();
        bool mine = false;
        for (int y = tid; y < h; y += PT) {          // row sweeps, both directions
            unsigned char* r = st + y * w;
            bool carry = false;
            for (int x = 0; x < w; ++x) {
                const int s_ = r[x];
                if (s_ == 0) carry = false;
                else if (s_ == 2) carry = true;
                else if (carry) { r[x] = 2; mine = true; }
            }
            carry = false;
            for (int x = w - 1; x >= 0; --x) {
                const int s_ = r[x];
                if (s_ == 0) carry = false;
                else if (s_ == 2) carry = true;
                else if (carry) { r[x] = 2; mine = true; }
            }
        }
        __syncthreads();
        for (int x = tid; x < w; x += PT) {          // column sweeps
            bool carry = false;
            for (int y = 0; y < h; ++y) {
                const int s_ = st[y * w + x];
                if (s_ == 0) carry = false;
                else if (s_ == 2) carry = true;
                else if (carry) { st[y * w + x] = 2; mine = true; }
            }
            carry = false;
            for (int y = h - 1; y >= 0; --y) {
                const int s_ = st[y * w + x];
                if (s_ == 0) carry = false;
                else if (s_ == 2) carry = true;
                else if (carry) { st[y * w + x] = 2; mine = true; }
            }
        }
        if (mine) *changed = 1;
        __syncthreads();
        const int again = *changed;
        __syncthreads();
        if (!again) break;
    }
    for (int i = tid; i < w * h; i += PT) {
        if (st[i] != 1) continue;
        const int y = i / w, x = i - y * w;
        const int64_t p = (int64_t)(y0 + y) * W + x0 + x;
        const int old = lab[p];
        if (old != 0) *hit = 1;      // the hole held another label: the reference's order matters from here on
        const int nv = old + v;
        lab[p] = nv;
        if (GROW && old != 0 && dirty) {      // labels `old` (it loses this pixel) and `old + v` (it gains it) are no longer the
            if (old < nmax) dirty[old] = 1;   // objects the frame's statistics describe: the replay has to look at them again
            if (nv < nmax) dirty[nv] = 1;
        }
        if (GROW && old != 0 && nv < nmax) {
            atomicMin(&bbox[4 * nv], x0 + x);
            atomicMin(&bbox[4 * nv + 1], y0 + y);
            atomicMax(&bbox[4 * nv + 2], x0 + x);
            atomicMax(&bbox[4 * nv + 3], y0 + y);
        }
    }
    __syncthreads();
    return 0;
}

__device__ __forceinline__ bool label_has_holes(const int32_t* e4, const int32_t* ncomp, int v) {
    const int q = e4[v] >= 0 ? e4[v] / 4 : -((-e4[v] + 3) / 4);      // floor division, as the host's e4 // 4
    return ncomp[v] - q > 0;
}

__global__ __launch_bounds__(PT) void fill_all_kernel(int32_t* lab, int H, int W, const int32_t* __restrict__ num_ptr,
                                                     int32_t* bbox, const int32_t* __restrict__ e4,
                                                     const int32_t* __restrict__ ncomp, int32_t* flags) {
    __shared__ unsigned char st[FILL_MAX_PIX];
    __shared__ int changed, hit;
    const int num = *num_ptr;
    if (threadIdx.x == 0) hit = 0;
    __syncthreads();
    for (int v = 1 + (int)blockIdx.x; v < num; v += (int)gridDim.x) {
        if (!label_has_holes(e4, ncomp, v)) continue;
        // (a crop beyond the LDS staging: left to the sequential replay, which floods it in global scratch)
        if (fill_one_object<false>(lab, H, W, v, bbox, 0, st, &changed, &hit) && threadIdx.x == 0) hit = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && hit) flags[0] = 1;
}

// Nested objects (flags[0] set by the concurrent pass above): the reference's order matters.  Restore the map from its
// snapshot (grid-wide) ...
__global__ void restore_if_dirty_kernel(int32_t* __restrict__ lab, const int32_t* __restrict__ snapshot, int64_t hw,
                                        const int32_t* __restrict__ flags) {
    if (flags[0] == 0) return;
    for (int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x; i < hw; i += (int64_t)gridDim.x * PT) lab[i] = snapshot[i];
}

// ... and replay Inference2D.py:80-91 in label order with ONE workgroup, still on the device: labels with holes (from the
// statistics), and -- once a hole has held another label -- also every label whose pixel set has CHANGED since the statistics
// were taken (`dirty`: label m loses the pixels that become m + v, label m + v gains them), each from the current map (bounding
// boxes grown as labels gain pixels).  Rounds 2-3 replayed EVERY label from the first nested hole on (32.7 ms for 310 objects at
// 832x992); but the reference's per-object fill only looks at bw = (labels == n): a label whose pixel set is unchanged and that
// had no hole when the statistics were taken has none now, so skipping it changes nothing.
// flags[2] = 1: the frame was replayed here; flags[1] = 1: a crop did not fit -- the host replays it.
__global__ __launch_bounds__(PT) void fill_sequential_kernel(int32_t* lab, int H, int W, const int32_t* __restrict__ num_ptr,
                                                            int32_t* bbox, const int32_t* __restrict__ e4,
                                                            const int32_t* __restrict__ ncomp, int nmax, int32_t* flags,
                                                            unsigned char* big, int64_t big_cap, int32_t* dirty) {
    __shared__ unsigned char st[FILL_MAX_PIX];
    __shared__ int changed, strict;
    if (flags[0] == 0 || flags[1] != 0) return;      // nothing nested -- or an oversize crop: left to the host
    const int num = *num_ptr < nmax ? *num_ptr : nmax;
    if (threadIdx.x == 0) strict = 0;
    for (int i = threadIdx.x; i < nmax; i += PT) dirty[i] = 0;
    __syncthreads();
    for (int v = 1; v < num; ++v) {
        // (uniform: `strict` and `dirty` are written before the barrier that ends fill_one_object)
        if (!label_has_holes(e4, ncomp, v) && !(strict != 0 && dirty[v] != 0)) continue;
        if (fill_one_object<true>(lab, H, W, v, bbox, nmax, st, &changed, &strict, big, big_cap, dirty)) {
            if (threadIdx.x == 0) flags[1] = 1;
            return;
        }
    }
    if (threadIdx.x == 0) flags[2] = 1;
}

// newid[v] = 1, 2, ... over the kept labels in label order (area inside [min_size, max_size], present in the field of view
// when a presence table is given), 0 for every other entry of the table; tail[0..3] = label count, dirty, oversize, replayed flags
// behind the uint16 map so that ONE device -> host copy carries the frame's result and its validity.
__global__ __launch_bounds__(1024) void newid_kernel(const int32_t* __restrict__ num_ptr, const int32_t* __restrict__ area,
                                                    const int32_t* __restrict__ present, int min_size, int max_size, int nmax,
                                                    int32_t* __restrict__ newid, const int32_t* __restrict__ flags,
                                                    int32_t* __restrict__ tail) {
    __shared__ int part[1024];
    __shared__ int base;
    const int num = *num_ptr < nmax ? *num_ptr : nmax;
    const int tid = threadIdx.x;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < nmax; start += 1024) {
        const int v = start + tid;
        const int keep = (v >= 1 && v < num && area[v] >= min_size && area[v] <= max_size && (!present || present[v])) ? 1 : 0;
        part[tid] = keep;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {       // inclusive scan (Hillis-Steele)
            const int add = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (v < nmax) newid[v] = keep ? base + part[tid] : 0;
        __syncthreads();
        if (tid == 1023) base += part[1023];
        __syncthreads();
    }
    if (tid == 0 && tail) {
        tail[0] = *num_ptr;
        tail[1] = flags[0];
        tail[2] = flags[1];
        tail[3] = flags[2];      // the nested-object replay ran on the device
    }
}

int run_ccl(const int32_t* val, int32_t* parent, int W, Rect r, int mode, int n, int nl, lu_stream_t stream) {
    const int64_t total = (int64_t)r.w * r.h;
    LU_LAUNCH(ccl_init_kernel, dim3(pgrid(total)), dim3(PT), stream, val, parent, W, r, mode, n, nl);
    LU_LAUNCH(ccl_merge_kernel, dim3(pgrid(total)), dim3(PT), stream, val, parent, W, r, mode, n, nl);
    LU_LAUNCH(ccl_flatten_kernel, dim3(pgrid(total)), dim3(PT), stream, parent, W, r);
    return LU_CHECK_LAUNCH();
}

}  // namespace

// Workspace layout (int32 words): cell | edge | parent | flag | key | cc   (H*W each)   mark | scanv (NB each)
extern "C" size_t lu_post_workspace_bytes(int32_t H, int32_t W) {
    const size_t hw = (size_t)H * W, nb = (size_t)((H + 1) / 2) * ((W + 1) / 2);
    return (6 * hw + 2 * nb + 16) * sizeof(int32_t);
}

extern "C" int32_t lu_post_max_labels(int32_t H, int32_t W) { return ((H + 1) / 2) * ((W + 1) / 2) + 2; }

extern "C" int lu_post_label(const float* softmax_chw, int32_t H, int32_t W, float edge_thresh, double edge_dist,
                             void* workspace, int32_t* labels, int32_t* num_labels, int32_t* area, lu_stream_t stream) {
    LU_REQUIRE(softmax_chw && workspace && labels && num_labels && area && H > 0 && W > 0 && edge_dist >= 0,
               "lu_post_label: bad arguments");
    LU_REQUIRE((int64_t)H * W < ((int64_t)1 << 30), "lu_post_label: frame too large");
    const int64_t hw = (int64_t)H * W;
    const int nb = ((H + 1) / 2) * ((W + 1) / 2);
    int32_t* ws = (int32_t*)workspace;
    int32_t *cell = ws, *edge = ws + hw, *parent = ws + 2 * hw, *flag = ws + 3 * hw, *key = ws + 4 * hw, *cc = ws + 5 * hw;
    int32_t *mark = ws + 6 * hw, *scanv = mark + nb;
    const Rect full{0, 0, W, H};
    const dim3 g(pgrid(hw)), b(PT);
    LU_LAUNCH(classify_kernel, g, b, stream, softmax_chw, hw, edge_thresh, cell, edge);
    // binary_fill_holes: background components (4-conn) that do not reach the frame border become cell
    if (run_ccl(cell, parent, W, full, MODE_BG4, 0, 0, stream)) return 1;
    LU_LAUNCH(fill32_kernel, g, b, stream, flag, 0, hw);
    LU_LAUNCH(border_flag_kernel, dim3(pgrid(2 * (int64_t)(W + H))), b, stream, (const int32_t*)parent, flag, W, full);
    LU_LAUNCH(fill_bg_kernel, g, b, stream, cell, (const int32_t*)parent, (const int32_t*)flag, edge, hw);
    // 8-connected components, numbered in OpenCV's order
    if (run_ccl(cell, parent, W, full, MODE_FG8, 0, 0, stream)) return 1;
    LU_LAUNCH(fill32_kernel, g, b, stream, key, 0x7fffffff, hw);
    LU_LAUNCH(fg_key_kernel, g, b, stream, (const int32_t*)parent, key, H, W);
    LU_LAUNCH(fill32_kernel, dim3(pgrid(nb)), b, stream, mark, 0, (int64_t)nb);
    LU_LAUNCH(key_mark_kernel, g, b, stream, (const int32_t*)parent, (const int32_t*)key, mark, hw);
    LU_LAUNCH(scan_kernel, dim3(1), dim3(1024), stream, (const int32_t*)mark, scanv, nb, num_labels);
    LU_LAUNCH(fill32_kernel, dim3(pgrid(nb + 2)), b, stream, area, 0, (int64_t)nb + 2);
    LU_LAUNCH(fg_assign_kernel, dim3(pgrid((hw + 7) / 8)), b, stream, (const int32_t*)parent, (const int32_t*)key, (const int32_t*)scanv, cc,
              area, hw);
    // edge pixels within edge_dist of a cell take the nearest cell's label
    int radius = 0;      // smallest window that holds every offset with sqrt(dy^2 + dx^2) < edge_dist
    while ((double)(radius + 1) < edge_dist) ++radius;
    LU_LAUNCH(absorb_kernel, g, b, stream, (const int32_t*)cc, (const int32_t*)edge, labels, H, W, edge_dist, radius);
    return LU_CHECK_LAUNCH();
}

/* bbox[4*v .. 4*v+3] = xmin, ymin, xmax, ymax; e4[v] = 4 * Euler number (8-connectivity); ncomp[v] = 8-connected components
 * of label v; for 1 <= v < num_labels (host value).  workspace as above. */
extern "C" int lu_post_label_stats(const int32_t* labels, int32_t H, int32_t W, int32_t num_labels, void* workspace,
                                   int32_t* bbox, int32_t* e4, int32_t* ncomp, lu_stream_t stream) {
    LU_REQUIRE(labels && workspace && bbox && e4 && ncomp && H > 0 && W > 0 && num_labels >= 1, "lu_post_label_stats: bad arguments");
    const int64_t hw = (int64_t)H * W;
    int32_t* parent = (int32_t*)workspace + 2 * hw;
    const Rect full{0, 0, W, H};
    const dim3 b(PT);
    LU_LAUNCH(fill32_kernel, dim3(pgrid(num_labels)), b, stream, e4, 0, (int64_t)num_labels);
    LU_LAUNCH(fill32_kernel, dim3(pgrid(num_labels)), b, stream, ncomp, 0, (int64_t)num_labels);
    LU_LAUNCH(bbox_init_kernel, dim3(pgrid(num_labels)), b, stream, bbox, (int64_t)num_labels);
    LU_LAUNCH(label_stats_kernel, dim3(pgrid((int64_t)(H + 1) * (W + 1))), b, stream, labels, H, W, num_labels, bbox, e4);
    if (run_ccl(labels, parent, W, full, MODE_EQ8, 0, num_labels, stream)) return 1;
    LU_LAUNCH(count_roots_kernel, dim3(pgrid(hw)), b, stream, (const int32_t*)parent, labels, ncomp, hw);
    return LU_CHECK_LAUNCH();
}

/* Holes of object n (labels == n) inside the crop [x0, x0+w) x [y0, y0+h) -- its bounding box grown by one pixel, clipped to
 * the frame: 4-connected components of "labels != n" that do not touch the crop border get labels += n (reference
 * Inference2D.py:80-91 through bbox_crop / binary_fill_holes / bbox_fill, additive quirk included).  *dirty (device int,
 * never cleared here) is set when such a hole held a non-zero label. */
extern "C" int lu_post_fill_object(int32_t* labels, int32_t H, int32_t W, int32_t n, int32_t x0, int32_t y0, int32_t w, int32_t h,
                                   void* workspace, int32_t* dirty, lu_stream_t stream) {
    LU_REQUIRE(labels && workspace && dirty && n > 0 && x0 >= 0 && y0 >= 0 && w > 0 && h > 0 && x0 + w <= W && y0 + h <= H,
               "lu_post_fill_object: bad arguments");
    const int64_t hw = (int64_t)H * W;
    int32_t *parent = (int32_t*)workspace + 2 * hw, *flag = (int32_t*)workspace + 3 * hw;
    const Rect r{x0, y0, w, h};
    const dim3 b(PT), g(pgrid((int64_t)w * h));
    if (run_ccl(labels, parent, W, r, MODE_NE4, n, 0, stream)) return 1;
    LU_LAUNCH(clear_flag_kernel, g, b, stream, (const int32_t*)parent, flag, W, r);
    LU_LAUNCH(border_flag_kernel, dim3(pgrid(2 * (int64_t)(w + h))), b, stream, (const int32_t*)parent, flag, W, r);
    LU_LAUNCH(object_fill_kernel, g, b, stream, labels, (const int32_t*)parent, (const int32_t*)flag, W, r, n, dirty);
    return LU_CHECK_LAUNCH();
}

/* Device-driven hole filling of every object at once (no host read between lu_post_label_stats and the final map):
 * num_labels / bbox / e4 / ncomp are the DEVICE arrays the two calls above filled.  flags[0] is set when a hole held a
 * non-zero label (nested objects: the reference's additive quirk makes its label order matter) or an object's crop exceeds
 * the kernel's LDS staging -- the result is then to be discarded: lu_post_frame restores the snapshot and replays
 * Inference2D.py:80-91 in label order on the device; a host that sequences the calls itself replays it object by object with
 * lu_post_fill_object.  flags is never cleared here. */
extern "C" int lu_post_fill_all(int32_t* labels, int32_t H, int32_t W, const int32_t* num_labels, const int32_t* bbox,
                                const int32_t* e4, const int32_t* ncomp, int32_t* flags, lu_stream_t stream) {
    LU_REQUIRE(labels && num_labels && bbox && e4 && ncomp && flags && H > 0 && W > 0, "lu_post_fill_all: bad arguments");
    int blocks = lu_post_max_labels(H, W);
    if (blocks > 2048) blocks = 2048;
    LU_LAUNCH(fill_all_kernel, dim3(blocks), dim3(PT), stream, labels, H, W, num_labels, (int32_t*)bbox, e4, ncomp, flags);
    return LU_CHECK_LAUNCH();
}

/* Size / field-of-view filter and consecutive numbering on the device (Inference2D.py:93-123): newid[v] for all
 * table_size entries (0 = dropped), from the DEVICE label count and areas; present may be NULL (no FOV mask).
 * tail (may be NULL) receives {label count, flags[0], flags[1]} -- placed behind the uint16 map by the caller so that one
 * copy brings back the frame. */
extern "C" int lu_post_newid(const int32_t* num_labels, const int32_t* area, const int32_t* present, int32_t min_size,
                             int32_t max_size, int32_t table_size, int32_t* newid, const int32_t* flags, int32_t* tail,
                             lu_stream_t stream) {
    LU_REQUIRE(num_labels && area && newid && flags && table_size >= 1, "lu_post_newid: bad arguments");
    LU_LAUNCH(newid_kernel, dim3(1), dim3(1024), stream, num_labels, area, present, min_size, max_size, table_size, newid,
              flags, tail);
    return LU_CHECK_LAUNCH();
}

/* One frame, one call: everything Inference2D.py:66-123 needs from the device, enqueued back to back on `stream` with no
 * host decision in between (the host thread pays ONE foreign call instead of ~12 calls and as many tensor-library copies;
 * at bf16 streaming rates -- 1.7 ms of forward per frame -- that host time is what bounds the frame rate).
 *   tables    : int32 [4 + 8 * lu_post_max_labels]: num | dirty | oversize | pad | area | bbox (4 per label) | e4 | ncomp | present
 *   snapshot  : int32 [H * W] copy of the label map taken before the hole filling (the exact fallback starts from it)
 *   out       : device, 4-byte aligned: uint16 map [H * W] padded to a multiple of 4 bytes, then int32 {num, dirty, oversize, 0}
 *   host_out  : the same bytes in PINNED host memory (asynchronous device -> host copy behind the kernels), or NULL
 * fov = 0: no field-of-view mask. */
extern "C" int lu_post_frame(const float* softmax_chw, int32_t H, int32_t W, float edge_thresh, double edge_dist,
                             int32_t min_size, int32_t max_size, int32_t fov, int32_t single_column, void* workspace,
                             int32_t* labels, int32_t* snapshot, int32_t* tables, int32_t* newid, void* out, void* host_out,
                             lu_stream_t stream) {
    LU_REQUIRE(softmax_chw && workspace && labels && snapshot && tables && newid && out && H > 0 && W > 0,
               "lu_post_frame: bad arguments");
    const int n = lu_post_max_labels(H, W);
    const int64_t hw = (int64_t)H * W;
    int32_t *num = tables, *flags = tables + 1, *area = tables + 4, *bbox = tables + 4 + n, *e4 = tables + 4 + 5 * n,
            *ncomp = tables + 4 + 6 * n, *present = tables + 4 + 7 * n;
    LU_LAUNCH(fill32_kernel, dim3(1), dim3(PT), stream, tables, 0, (int64_t)4);
    if (lu_post_label(softmax_chw, H, W, edge_thresh, edge_dist, workspace, labels, num, area, stream)) return 1;
    if (lu_post_label_stats(labels, H, W, n, workspace, bbox, e4, ncomp, stream)) return 1;
    LU_LAUNCH(copy32_kernel, dim3(pgrid(hw)), dim3(PT), stream, (const int32_t*)labels, snapshot, hw);
    if (lu_post_fill_all(labels, H, W, num, bbox, e4, ncomp, flags, stream)) return 1;
    // nested objects: exact replay in label order, still on the device (both launches return at once when nothing is nested)
    LU_LAUNCH(restore_if_dirty_kernel, dim3(pgrid(hw)), dim3(PT), stream, labels, (const int32_t*)snapshot, hw, (const int32_t*)flags);
    // (global scratch for crops beyond the LDS staging: the `flag` plane of the workspace, 4 H W bytes, idle at this point)
    LU_LAUNCH(fill_sequential_kernel, dim3(1), dim3(PT), stream, labels, H, W, (const int32_t*)num, bbox, (const int32_t*)e4,
              (const int32_t*)ncomp, n, flags, (unsigned char*)((int32_t*)workspace + 3 * hw), (int64_t)4 * hw,
              present /* = the replay's `dirty` table: idle until the tail below recomputes the presence table */);
    return lu_post_frame_tail(H, W, min_size, max_size, fov, single_column, labels, tables, newid, out, host_out, stream);
}

/* The part of lu_post_frame behind the hole filling (presence table, numbering, relabel, copy to the host) -- also what the
 * host runs after its exact sequential replay of a nested-object frame. */
extern "C" int lu_post_frame_tail(int32_t H, int32_t W, int32_t min_size, int32_t max_size, int32_t fov, int32_t single_column,
                                  const int32_t* labels, int32_t* tables, int32_t* newid, void* out, void* host_out,
                                  lu_stream_t stream) {
    LU_REQUIRE(labels && tables && newid && out && H > 0 && W > 0, "lu_post_frame_tail: bad arguments");
    const int n = lu_post_max_labels(H, W);
    const int64_t hw = (int64_t)H * W;
    const int64_t map_words = (hw + 1) / 2;
    int32_t *num = tables, *flags = tables + 1, *area = tables + 4, *present = tables + 4 + 7 * n;
    if (fov && lu_post_present(labels, H, W, fov, single_column, n, present, stream)) return 1;
    if (lu_post_newid(num, area, fov ? present : nullptr, min_size, max_size, n, newid, flags, (int32_t*)out + map_words, stream))
        return 1;
    if (lu_post_relabel(labels, H, W, newid, n, (uint16_t*)out, stream)) return 1;
    if (host_out) {
#ifdef LU_EMU
        memcpy(host_out, out, (size_t)(map_words + 4) * 4);
#else
        if (hipMemcpyAsync(host_out, out, (size_t)(map_words + 4) * 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) {
            lu_set_error("lu_post_frame_tail: device -> host copy failed");
            return 1;
        }
#endif
    }
    return LU_CHECK_LAUNCH();
}

/* box[0..3] = xmin, ymin, xmax, ymax of labels == n (xmin = INT_MAX when the label is absent); the strictly sequential
 * fall-back of the hole filling needs the box of the CURRENT label set */
extern "C" int lu_post_bbox_of_label(const int32_t* labels, int32_t H, int32_t W, int32_t n, int32_t* box, lu_stream_t stream) {
    LU_REQUIRE(labels && box && H > 0 && W > 0, "lu_post_bbox_of_label: bad arguments");
    LU_LAUNCH(bbox_init_kernel, dim3(1), dim3(PT), stream, box, (int64_t)1);
    LU_LAUNCH(bbox_of_label_kernel, dim3(pgrid((int64_t)H * W)), dim3(PT), stream, labels, H, W, n, box);
    return LU_CHECK_LAUNCH();
}

/* present[v] = 1 for every label value v in [0, num_labels) that occurs inside the field of view (rows [fov, H-fov),
 * columns [.., W-fov); left side: single_column = 1 masks only column `fov` -- the reference's `fov_im[:, FOV] = 0`,
 * Inference2D.py:97 -- 0 masks columns [0, fov)); masked pixels count as label 0, as `labels * fov_im` does. */
extern "C" int lu_post_present(const int32_t* labels, int32_t H, int32_t W, int32_t fov, int32_t single_column, int32_t num_labels,
                               int32_t* present, lu_stream_t stream) {
    LU_REQUIRE(labels && present && H > 0 && W > 0 && fov >= 0 && num_labels >= 1, "lu_post_present: bad arguments");
    LU_LAUNCH(fill32_kernel, dim3(pgrid(num_labels)), dim3(PT), stream, present, 0, (int64_t)num_labels);
    LU_LAUNCH(present_kernel, dim3(pgrid((int64_t)H * W)), dim3(PT), stream, labels, H, W, fov, single_column, num_labels, present);
    return LU_CHECK_LAUNCH();
}

/* out[p] = newid[labels[p]] for 0 < labels[p] < num_labels, else 0 (uint16): the consecutive relabelling of
 * Inference2D.py:113-123 with newid = 0 for filtered labels */
extern "C" int lu_post_relabel(const int32_t* labels, int32_t H, int32_t W, const int32_t* newid, int32_t num_labels, uint16_t* out,
                               lu_stream_t stream) {
    LU_REQUIRE(labels && newid && out && H > 0 && W > 0 && num_labels >= 1, "lu_post_relabel: bad arguments");
    LU_LAUNCH(relabel_kernel, dim3(pgrid((int64_t)H * W)), dim3(PT), stream, labels, newid, num_labels, (unsigned short*)out,
              (int64_t)H * W);
    return LU_CHECK_LAUNCH();
}
