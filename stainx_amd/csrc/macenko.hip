// Macenko stain normalisation for MI355X (gfx950) -- hand-written HIP, wave64.
//
// What it computes is MacenkoTorch.transform / compute_reference_stain_matrix_torch of
// rendeirolab/stainx (src/stainx/backends/torch_backend.py:399-560).  How it computes it is this
// library's own design:
//
//   * every pass over the pixels is one streaming kernel (planar NCHW, 16-byte loads per lane,
//     256-thread workgroups, N*ceil(P/8192) workgroups so the 256 CUs are oversubscribed);
//   * the four nearest-rank order statistics per tile (phi@1%, phi@99%, C0@99%, C1@99%;
//     torch_backend.py:363-365) are EXACT but never sort: a 4096-pixel strided sample of the tile is
//     sorted in LDS to get a bracket [lo,hi] around each wanted rank; the streaming pass counts the
//     pixels below the bracket (integer adds) and gathers the few keys inside it; one workgroup per
//     tile then radix-selects the wanted rank among those candidates.  If a bracket misses or
//     overflows (heavy ties, adversarial data) that workgroup radix-selects over the whole tile
//     instead -- slower, same answer;
//   * the 3x3 covariance is accumulated in fp64 (raw moments cancel catastrophically in fp32) and
//     diagonalised by cyclic Jacobi in fp64; eigenvector signs follow the "positive component sum"
//     convention (the transform is invariant to them on real H&E tiles).
//
// Launch sequence of one transform:  stats -> plane -> angle pass -> stain vectors -> concentration
// pass -> scale -> reconstruct.  All stream-ordered, no host synchronisation.
#include "common.hpp"

namespace sx {
namespace macenko {

#define SX_STAMP(st, i) do { if (threadIdx.x == 0) (st).stamp[i] = wall_clock64(); } while (0)

constexpr int kSample = 4096;          // sorted sample per tile (LDS bitonic)
constexpr int kCap = 32768;            // candidate keys per selection slot
constexpr int kGroupThreads = 1024;    // per-tile scalar kernels
constexpr int kIters = 8;              // pixel packs per lane per streaming workgroup
constexpr int kSlots = 4;              // 0: phi@1  1: phi@99  2: C0@99  3: C1@99
constexpr int kMoments = 20;           // [cnt,sx,sy,sz,xx,xy,xz,yy,yz,zz] masked, then all pixels

constexpr float kBeta = 0.15f;         // torch_backend.py:542
constexpr float kIo = 240.0f;          // torch_backend.py:541
constexpr float kLn2 = 0.693147180559945309f;
constexpr float kLnIo = 5.48063892334199f;       // ln 240
constexpr float kLog2e = 1.44269504088896341f;

struct alignas(256) GroupState {
    double mom[kMoments];
    double cov[9];
    float vecs[6];      // (3,2) row-major, columns [middle, largest] eigenvalue
    float he[6];        // (3,2) row-major HE_source
    float pinv[6];      // (2,3) row-major pseudo-inverse of HE_source
    float phi[2];
    float max_c[2];
    float scale[2];     // target_max_conc / max_c
    unsigned long long n_sel;     // pixels in the selection set (kept by the OD filter, or all)
    unsigned long long rank[kSlots];   // wanted 0-based rank inside the selection set
    uint32_t lo_key[kSlots], hi_key[kSlots];
    double bin_origin[kSlots], bin_scale[kSlots];   // bracket-relative bin of a candidate: (value - origin) * scale
    uint32_t hist_bad;            // bit s: a workgroup's LDS queue spilled, block histograms of slot s are incomplete
    uint32_t below[kSlots];       // keys < lo_key        (atomic, integer => order independent)
    uint32_t ncand[kSlots];       // keys in [lo_key,hi_key] (atomic)
    uint32_t ncand_seen[kSlots];  // copy kept for sx_macenko_tile_params
    int32_t use_all;
    uint32_t fell_back;           // bit s: slot s used the full-tile radix select
    unsigned long long stamp[16]; // diagnostic: wall_clock64() at stage boundaries of the per-tile kernels
};

struct Geometry {
    int64_t n_tiles, pixels;      // P = H*W
    int blocks_per_tile;          // streaming workgroups per tile
    int chunk;                    // pixels per streaming workgroup
    int pooled;                   // 1: one group over all tiles (fit), 0: one group per tile
};

struct Workspace {
    GroupState* state;
    double* partial;              // [n_tiles*blocks_per_tile][kMoments]
    uint32_t* cand;               // [groups][kSlots][kCap]
    uint32_t* block_hist;         // [n_tiles*blocks_per_tile][2][256] bracket-relative histograms of a pass's candidates
    float* sample_od;             // [groups][3][kSample] OD of the strided sample (written once, read by both per-tile kernels)
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int blocks_per_tile_for(int64_t pixels) {
    const int64_t chunk = (int64_t)kStreamThreads * 4 * kIters;   // sized for the vector path; the scalar path uses the same count
    return (int)((pixels + chunk - 1) / chunk);
}

static size_t workspace_bytes(int64_t n_tiles, int64_t pixels) {
    const size_t b = (size_t)blocks_per_tile_for(pixels);
    size_t total = align_up(sizeof(GroupState) * (size_t)n_tiles, 256);
    total += align_up(sizeof(double) * kMoments * b * (size_t)n_tiles, 256);
    total += align_up(sizeof(uint32_t) * kSlots * kCap * (size_t)n_tiles, 256);
    total += align_up(sizeof(uint32_t) * 2 * 256 * b * (size_t)n_tiles, 256);
    total += align_up(sizeof(float) * 3 * kSample * (size_t)n_tiles, 256);
    return total;
}

static Workspace carve(void* base, int64_t n_tiles, int64_t pixels) {
    Workspace w;
    char* p = static_cast<char*>(base);
    const size_t b = (size_t)blocks_per_tile_for(pixels);
    w.state = reinterpret_cast<GroupState*>(p);
    p += align_up(sizeof(GroupState) * (size_t)n_tiles, 256);
    w.partial = reinterpret_cast<double*>(p);
    p += align_up(sizeof(double) * kMoments * b * (size_t)n_tiles, 256);
    w.cand = reinterpret_cast<uint32_t*>(p);
    p += align_up(sizeof(uint32_t) * kSlots * kCap * (size_t)n_tiles, 256);
    w.block_hist = reinterpret_cast<uint32_t*>(p);
    p += align_up(sizeof(uint32_t) * 2 * 256 * b * (size_t)n_tiles, 256);
    w.sample_od = reinterpret_cast<float*>(p);
    return w;
}

// ------------------------------------------------------------------------------------------------
// per-pixel arithmetic
// ------------------------------------------------------------------------------------------------
// OD = -log((x*255+1)/240) (torch_backend.py:550) evaluated as ln240 - ln2*log2(x*255+1):
// one fma, v_log_f32, one fma.  Differs from the reference's mul/add/div/log chain by ~1e-7 absolute.
__device__ __forceinline__ float optical_density(float unit) {
    const float t = fmaf(unit, 255.0f, 1.0f);
    return fmaf(-kLn2, __log2f(t), kLnIo);
}

__device__ __forceinline__ bool od_selected(const float od[3], bool use_all) {
    return use_all || (fminf(od[0], fminf(od[1], od[2])) >= kBeta);    // torch_backend.py:404-405
}

__device__ __forceinline__ uint32_t angle_key(const float od[3], const float* __restrict__ v) {
    const float t0 = fmaf(od[2], v[4], fmaf(od[1], v[2], od[0] * v[0]));   // That[:,0]  (:417)
    const float t1 = fmaf(od[2], v[5], fmaf(od[1], v[3], od[0] * v[1]));   // That[:,1]
    return float_key(atan2f(t1, t0));                                       // :418
}

__device__ __forceinline__ void concentration(const float od[3], const float* __restrict__ pinv, float& c0, float& c1) {
    c0 = fmaf(od[2], pinv[2], fmaf(od[1], pinv[1], od[0] * pinv[0]));     // :444
    c1 = fmaf(od[2], pinv[5], fmaf(od[1], pinv[4], od[0] * pinv[3]));
}

template <typename T>
__device__ __forceinline__ void load_od_scalar(const T* __restrict__ images, int64_t pixels, int64_t tile, int64_t p, float od[3]) {
    const T* base = images + tile * 3 * pixels + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) od[c] = optical_density(Elem<T>::load(base[c * pixels]));
}

// ------------------------------------------------------------------------------------------------
// pass 1: raw moments of the OD vectors
// ------------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(kStreamThreads) void stats_kernel(const T* __restrict__ images, Geometry g, double* __restrict__ partial) {
    const int64_t tile = blockIdx.x / g.blocks_per_tile;
    const int chunk_id = blockIdx.x % g.blocks_per_tile;
    const int64_t p_begin = (int64_t)chunk_id * g.chunk;
    const int64_t p_end = min(p_begin + g.chunk, g.pixels);
    const T* img = images + tile * 3 * g.pixels;

    double acc[kMoments];
#pragma unroll
    for (int k = 0; k < kMoments; ++k) acc[k] = 0.0;

    for (int64_t p = p_begin + (int64_t)threadIdx.x * V; p < p_end; p += (int64_t)kStreamThreads * V) {
        float u[3][V];
#pragma unroll
        for (int c = 0; c < 3; ++c) load_unit<T, V>(img + c * g.pixels + p, u[c]);
        // fp32 sums over the V pixels of this pack, fp64 across packs / lanes / workgroups
        float m[10], a[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) m[k] = a[k] = 0.0f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float od[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) od[c] = optical_density(u[c][i]);
            const float keep = od_selected(od, false) ? 1.0f : 0.0f;
            const float pr[10] = {1.0f, od[0], od[1], od[2], od[0] * od[0], od[0] * od[1], od[0] * od[2], od[1] * od[1], od[1] * od[2], od[2] * od[2]};
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                a[k] += pr[k];
                m[k] = fmaf(keep, pr[k], m[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            acc[k] += (double)m[k];
            acc[10 + k] += (double)a[k];
        }
    }

    __shared__ double red[kStreamThreads / kWave][kMoments];
    const int wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < kMoments; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane_id() == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < kMoments) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kStreamThreads / kWave; ++w) s += red[w][threadIdx.x];
        partial[(int64_t)blockIdx.x * kMoments + threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// helpers of the per-tile kernels
// ------------------------------------------------------------------------------------------------
// One Jacobi rotation in the (p,q) plane of a symmetric 3x3 kept in scalars (r is the third index).
#define SX_JACOBI_ROTATE(app, aqq, apq, arp, arq, v0p, v0q, v1p, v1q, v2p, v2q)            \
    if ((apq) != 0.0) {                                                                      \
        const double theta = ((aqq) - (app)) / (2.0 * (apq));                                \
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;                                \
        (app) -= t * (apq);                                                                  \
        (aqq) += t * (apq);                                                                  \
        (apq) = 0.0;                                                                         \
        const double rp = (arp), rq = (arq);                                                 \
        (arp) = c * rp - sn * rq;                                                            \
        (arq) = sn * rp + c * rq;                                                            \
        double xp = (v0p), xq = (v0q);                                                       \
        (v0p) = c * xp - sn * xq;                                                            \
        (v0q) = sn * xp + c * xq;                                                            \
        xp = (v1p), xq = (v1q);                                                              \
        (v1p) = c * xp - sn * xq;                                                            \
        (v1q) = sn * xp + c * xq;                                                            \
        xp = (v2p), xq = (v2q);                                                              \
        (v2p) = c * xp - sn * xq;                                                            \
        (v2q) = sn * xp + c * xq;                                                            \
    }

// Eigen-decomposition of a symmetric 3x3 by cyclic Jacobi rotations (fp64, everything in registers).
// Eigenvalues ascending in w[], matching eigenvectors in the columns of q (row-major 3x3).
__device__ void jacobi_eigh3(const double a_in[9], double w[3], double q[9]) {
    double a00 = a_in[0], a01 = a_in[1], a02 = a_in[2], a11 = a_in[4], a12 = a_in[5], a22 = a_in[8];
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = a01 * a01 + a02 * a02 + a12 * a12;
        const double diag = a00 * a00 + a11 * a11 + a22 * a22;
        if (off <= 1e-36 * diag || off == 0.0) break;
        SX_JACOBI_ROTATE(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21)   // (0,1), r = 2
        SX_JACOBI_ROTATE(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22)   // (0,2), r = 1
        SX_JACOBI_ROTATE(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22)   // (1,2), r = 0
    }
    double d0 = a00, d1 = a11, d2 = a22;
    // stable ascending sort of (eigenvalue, column): bubble network with strict comparisons
#define SX_SWAP_COLS(da, db, xa0, xa1, xa2, xb0, xb1, xb2) \
    if ((da) > (db)) {                                     \
        double t_ = (da); (da) = (db); (db) = t_;          \
        t_ = (xa0); (xa0) = (xb0); (xb0) = t_;             \
        t_ = (xa1); (xa1) = (xb1); (xb1) = t_;             \
        t_ = (xa2); (xa2) = (xb2); (xb2) = t_;             \
    }
    SX_SWAP_COLS(d0, d1, v00, v10, v20, v01, v11, v21)
    SX_SWAP_COLS(d1, d2, v01, v11, v21, v02, v12, v22)
    SX_SWAP_COLS(d0, d1, v00, v10, v20, v01, v11, v21)
#undef SX_SWAP_COLS
    w[0] = d0; w[1] = d1; w[2] = d2;
    q[0] = v00; q[1] = v01; q[2] = v02;
    q[3] = v10; q[4] = v11; q[5] = v12;
    q[6] = v20; q[7] = v21; q[8] = v22;
}
#undef SX_JACOBI_ROTATE

// Angle percentiles -> extreme stain vectors -> HE_source (H before E) -> its (2,3) pseudo-inverse.
__device__ void stain_vectors_and_pinv(const float* vecs, float phi_lo, float phi_hi, float* he_out, float* pinv_out) {
    const float cl = cosf(phi_lo), sl = sinf(phi_lo), ch = cosf(phi_hi), sh = sinf(phi_hi);   // torch_backend.py:427-430
    float vmin[3], vmax[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        vmin[r] = fmaf(vecs[r * 2 + 1], sl, vecs[r * 2] * cl);                                   // :436
        vmax[r] = fmaf(vecs[r * 2 + 1], sh, vecs[r * 2] * ch);                                   // :437
    }
    const bool min_first = vmin[0] > vmax[0];                                                    // :439
    float he[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        he[r * 2] = min_first ? vmin[r] : vmax[r];
        he[r * 2 + 1] = min_first ? vmax[r] : vmin[r];
    }
    // pseudo-inverse (2,3) of HE (3,2) in fp64 through the eigen-decomposition of HE^T HE, dropping a
    // singular value below 3*eps_f32 of the largest (rank rule of lstsq(rcond=None), torch_backend.py:379)
    const double a = (double)he[0] * he[0] + (double)he[2] * he[2] + (double)he[4] * he[4];
    const double b = (double)he[0] * he[1] + (double)he[2] * he[3] + (double)he[4] * he[5];
    const double d = (double)he[1] * he[1] + (double)he[3] * he[3] + (double)he[5] * he[5];
    const double tr = a + d, df = a - d;
    const double disc = sqrt(df * df + 4.0 * b * b);
    const double l1 = 0.5 * (tr + disc), l2 = 0.5 * (tr - disc);
    double e1x, e1y;                                   // unit eigenvector of l1
    if (fabs(b) > 0.0) {
        e1x = l1 - d;
        e1y = b;
    } else if (a >= d) {
        e1x = 1.0;
        e1y = 0.0;
    } else {
        e1x = 0.0;
        e1y = 1.0;
    }
    const double nrm = sqrt(e1x * e1x + e1y * e1y);
    e1x /= nrm;
    e1y /= nrm;
    const double e2x = -e1y, e2y = e1x;
    const double rc = 3.0 * 1.1920928955078125e-07;
    const double i1 = l1 > 0.0 ? 1.0 / l1 : 0.0;
    const double i2 = (l2 > 0.0 && sqrt(l2) > rc * sqrt(l1)) ? 1.0 / l2 : 0.0;
    // (HE^T HE)^+ = i1 e1 e1^T + i2 e2 e2^T
    const double g00 = i1 * e1x * e1x + i2 * e2x * e2x, g01 = i1 * e1x * e1y + i2 * e2x * e2y, g11 = i1 * e1y * e1y + i2 * e2y * e2y;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pinv_out[c] = (float)(g00 * he[c * 2] + g01 * he[c * 2 + 1]);
        pinv_out[3 + c] = (float)(g01 * he[c * 2] + g11 * he[c * 2 + 1]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) he_out[i] = he[i];
}

// k = round(0.01*q*(n-1)), half to even, evaluated in double like the Python expression at
// torch_backend.py:364 (0-based rank).
__device__ __forceinline__ unsigned long long nearest_rank_index(double q, unsigned long long n) {
    if (n == 0) return 0;
    return (unsigned long long)rint((0.01 * q) * (double)(n - 1));
}

// Sample ranks bracketing the wanted rank: +-6 standard deviations of the rank a sample of m_valid keys
// gives to the k0-th of n_total keys, plus slack.  A miss is detected later and repaired.
__device__ __forceinline__ void bracket_ranks(int m_valid, unsigned long long n_total, unsigned long long k0, long long& lo_r, long long& hi_r) {
    const double f = n_total > 1 ? (double)k0 / (double)(n_total - 1) : 0.0;
    const double r = f * (double)(m_valid - 1);
    const double sd = sqrt((double)m_valid * f * (1.0 - f));
    lo_r = (long long)floor(r - 6.0 * sd - 3.0);
    hi_r = (long long)ceil(r + 6.0 * sd + 3.0);
}

// One wave turns a 256-bin histogram and a rank into (digit, rank inside that digit's bin).
__device__ __forceinline__ void scan_pick(const uint32_t* hist, unsigned long long rank, uint32_t& digit, unsigned long long& rank_in_bin) {
    const int lane = (int)lane_id();
    const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const unsigned long long mine = (unsigned long long)h0 + h1 + h2 + h3;
    unsigned long long incl = mine;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const unsigned long long up = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += up;
    }
    const uint64_t over = __ballot(incl > rank);
    const int owner = over ? (__ffsll((long long)over) - 1) : (kWave - 1);
    const unsigned long long before = __shfl(incl - mine, owner, kWave);
    const uint32_t b0 = __shfl(h0, owner, kWave), b1 = __shfl(h1, owner, kWave), b2 = __shfl(h2, owner, kWave);
    unsigned long long r = rank - before;
    uint32_t d = 0;
    if (r >= b0) { r -= b0; d = 1; if (r >= b1) { r -= b1; d = 2; if (r >= b2) { r -= b2; d = 3; } } }
    digit = 4u * (uint32_t)owner + d;
    rank_in_bin = r;
}

// Bracket-relative bin (0..255) of a key in [lo,hi], linear in the float VALUE the key stands for (key-linear
// bins would crowd: float keys spend one binade per exponent) and monotone in the key.  scale = 256/(v_hi-v_lo);
// an open or degenerate bracket gets scale 0 (everything in bin 0 -> the radix paths take over).
__device__ __forceinline__ uint32_t bin_of(uint32_t key, double lo_value, double scale) {
    const double d = ((double)key_float(key) - lo_value) * scale;
    return (uint32_t)fmin(fmax(d, 0.0), 255.0);     // values beyond the range (open brackets) go to the end bins
}
__device__ __forceinline__ double bin_origin_for(uint32_t lo) { return (double)key_float(lo); }
__device__ __forceinline__ double bin_scale_for(uint32_t lo, uint32_t hi) {
    const double span = (double)key_float(hi) - (double)key_float(lo);
    return (span > 0.0 && span < 1e300) ? 256.0 / span : 0.0;
}

// Exact r-th smallest (0-based, ties by index) of a short LDS list by rank counting; the owner writes *out.
__device__ __forceinline__ void rank_pick(const uint32_t* list, uint32_t n, unsigned long long r, uint32_t* out) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t k = list[i];
        uint32_t c = 0;
#pragma unroll 16
        for (uint32_t t = 0; t < n; ++t) {
            const uint32_t x = list[t];
            c += (x < k || (x == k && t < i)) ? 1u : 0u;
        }
        if (c == r) *out = k;
    }
}

struct SelectJob {
    uint32_t count;              // 0: nothing to select
    unsigned long long rank;     // 0-based, among the keys that lie in [lo,hi]
    uint32_t lo, hi;             // keys outside [lo,hi] are ignored
};

template <int R> struct SelectShared {
    uint32_t hist[R][256];
    unsigned long long rank[R];
    uint32_t digit[R];
};

// R exact order statistics at once (whole workgroup, >= R waves) over keys held in REGISTERS: every
// thread owns up to KPT keys of each of NSETS key sets, job r selects in set set_of[r].  Radix rounds of
// up to 8 bits over key - lo, starting at the top bit of hi - lo, so the first round already spreads
// the keys over the whole LDS histogram (byte-aligned digits of clustered float keys would pile onto a
// few bins and serialise the LDS atomics).  One histogram per job; wave j resolves job j.
template <int R, int NSETS, int KPT>
__device__ void multi_select(const uint32_t (&keys)[NSETS][KPT], const int (&n_mine)[NSETS], const int (&set_of)[R], const SelectJob (&job)[R],
                             uint32_t (&result)[R], SelectShared<R>* sh) {
    unsigned long long found[R];   // value of (key - lo) >> s_prev fixed so far
    int s_prev[R];                 // bits of key - lo still undetermined
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t span = job[r].hi - job[r].lo;
        found[r] = 0;
        s_prev[r] = (job[r].count == 0 || span == 0) ? 0 : 32 - __clz(span);
        if (threadIdx.x == 0) sh->rank[r] = job[r].rank;
    }
    const int wave = threadIdx.x / kWave;
    for (;;) {
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; ++r) any |= s_prev[r] > 0;
        if (!any) break;
        for (int t = threadIdx.x; t < R * 256; t += blockDim.x) (&sh->hist[0][0])[t] = 0;
        __syncthreads();
        int s_now[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s_now[r] = s_prev[r] > 8 ? s_prev[r] - 8 : 0;
            if (s_prev[r] <= 0) continue;
            const uint32_t lo = job[r].lo, hi = job[r].hi;
            const unsigned long long base = found[r] << (s_prev[r] - s_now[r]);
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t k = keys[set_of[r]][i];
                if (i >= n_mine[set_of[r]] || k < lo || k > hi) continue;
                const unsigned long long kp = (unsigned long long)(k - lo);
                if ((kp >> s_prev[r]) == found[r]) atomicAdd(&sh->hist[r][(uint32_t)((kp >> s_now[r]) - base)], 1u);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (wave == r && s_prev[r] > 0) {
                uint32_t d;
                unsigned long long rb;
                scan_pick(sh->hist[r], sh->rank[r], d, rb);
                if (lane_id() == 0) {
                    sh->digit[r] = d;
                    sh->rank[r] = rb;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (s_prev[r] <= 0) continue;
            found[r] = (found[r] << (s_prev[r] - s_now[r])) + sh->digit[r];
            s_prev[r] = s_now[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) result[r] = job[r].lo + (uint32_t)found[r];
}

// Slow path: exact rank-th smallest of the valid keys produced by key_at(i), i in [0,count), recomputed
// from the pixels in every round.  Whole workgroup.
template <class KeyAt>
__device__ uint32_t radix_select_stream(unsigned long long count, unsigned long long rank, KeyAt key_at, SelectShared<1>* sh) {
    uint32_t prefix = 0, mask = 0;
    if (threadIdx.x == 0) sh->rank[0] = rank;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int t = threadIdx.x; t < 256; t += blockDim.x) sh->hist[0][t] = 0;
        __syncthreads();
        for (unsigned long long i = threadIdx.x; i < count; i += blockDim.x) {
            uint32_t k;
            if (key_at(i, k) && ((k ^ prefix) & mask) == 0) atomicAdd(&sh->hist[0][(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < kWave) {
            uint32_t d;
            unsigned long long rb;
            scan_pick(sh->hist[0], sh->rank[0], d, rb);
            if (lane_id() == 0) {
                sh->digit[0] = d;
                sh->rank[0] = rb;
            }
        }
        __syncthreads();
        prefix |= sh->digit[0] << shift;
        mask |= 0xFFu << shift;
    }
    return prefix;
}

// Pixel walker of one group: tile-local for transform, all tiles for the pooled fit.
struct GroupPixels {
    int64_t first_tile, pixels, count;   // count = pixels in the group
    int pooled;
    __device__ __forceinline__ void locate(int64_t i, int64_t& tile, int64_t& p) const {
        if (!pooled) {       // no 64-bit division on the per-tile path
            tile = first_tile;
            p = i;
        } else {
            tile = i / pixels;
            p = i - tile * pixels;
        }
    }
};

__device__ __forceinline__ GroupPixels group_pixels(const Geometry& g, int group) {
    GroupPixels gp;
    gp.pixels = g.pixels;
    gp.pooled = g.pooled;
    gp.first_tile = g.pooled ? 0 : group;
    gp.count = g.pooled ? g.n_tiles * g.pixels : g.pixels;
    return gp;
}

// j-th of m evenly spaced sample positions in [0,count): floor(j*count/m); m is kSample (a power of two)
// whenever count >= kSample, so the division is a shift there.
__device__ __forceinline__ int64_t sample_position(int j, int m, int64_t count) {
    const unsigned long long prod = (unsigned long long)j * (unsigned long long)count;
    return (int64_t)(m == kSample ? prod / (unsigned)kSample : prod / (unsigned long long)m);
}

constexpr int kSamplePerThread = kSample / kGroupThreads;   // 4 sample keys per thread, kept in registers

constexpr int kShortList = 1024;   // keys of one histogram bin gathered for rank counting

struct SampleShared {
    union {
        SelectShared<4> sel;                 // radix fallback
        struct {
            uint32_t hist[2][256];           // one histogram per key set
            uint32_t list[4][kShortList];    // keys of the four picked bins
        } two;
    };
    uint32_t lo[2], hi[2];
    uint32_t bin[4], count[4], result[4];
    unsigned long long rank_in_bin[4];
    int valid;
};

// Brackets for two wanted ranks (k0[0], k0[1]) in key sets A and B from the rank statistics of a
// kSample-key sample spread over the workgroup's registers (invalid entries are 0xFFFFFFFF).
// Four sample order statistics (two ranks per key set) in two levels: one 256-bin histogram per key set over
// [min,max] (a single round of LDS atomics), one wave per rank picks its bin, the keys of that bin
// are gathered and the exact element is found by rank counting.  A crowded bin (> kShortList keys)
// falls back to the radix rounds of multi_select.
template <int NSETS>
__device__ void sample_brackets(const uint32_t (&keys)[NSETS][kSamplePerThread], unsigned long long n_total, const unsigned long long (&k0)[2], uint32_t (&lo)[2],
                                uint32_t (&hi)[2], double (&bin_origin)[2], double (&bin_scale)[2], SampleShared* sh) {
    if (threadIdx.x < 2) {
        sh->lo[threadIdx.x] = 0xFFFFFFFFu;
        sh->hi[threadIdx.x] = 0u;
    }
    if (threadIdx.x < 4) sh->count[threadIdx.x] = 0;
    if (threadIdx.x == 0) sh->valid = 0;
    for (int t = threadIdx.x; t < 2 * 256; t += blockDim.x) (&sh->two.hist[0][0])[t] = 0;
    __syncthreads();
    int valid = 0;
#pragma unroll
    for (int set = 0; set < NSETS; ++set) {
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
        for (int i = 0; i < kSamplePerThread; ++i) {
            const uint32_t k = keys[set][i];
            if (k != 0xFFFFFFFFu) {
                mn = min(mn, k);
                mx = max(mx, k);
                if (set == 0) ++valid;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn = min(mn, (uint32_t)__shfl_down(mn, off, kWave));
            mx = max(mx, (uint32_t)__shfl_down(mx, off, kWave));
        }
        if (lane_id() == 0) {
            atomicMin(&sh->lo[set], mn);
            atomicMax(&sh->hi[set], mx);
        }
    }
    valid = (int)wave_sum_u32((uint32_t)valid);
    if (lane_id() == 0 && valid) atomicAdd(&sh->valid, valid);
    __syncthreads();
    const int m_valid = sh->valid;
    long long lo_r[2], hi_r[2];
    bool want[4];
    unsigned long long rank[4];
    uint32_t set_lo[NSETS], set_hi[NSETS];
    double origin[NSETS], scale[NSETS];
#pragma unroll
    for (int set = 0; set < NSETS; ++set) {
        set_lo[set] = sh->lo[set];
        set_hi[set] = sh->hi[set];
        origin[set] = bin_origin_for(set_lo[set]);
        scale[set] = bin_scale_for(set_lo[set], set_hi[set]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bracket_ranks(m_valid, n_total, k0[s], lo_r[s], hi_r[s]);
        want[2 * s] = m_valid > 0 && lo_r[s] >= 0;
        want[2 * s + 1] = m_valid > 0 && hi_r[s] < m_valid;
        rank[2 * s] = (unsigned long long)(lo_r[s] < 0 ? 0 : lo_r[s]);
        rank[2 * s + 1] = (unsigned long long)(hi_r[s] < 0 ? 0 : hi_r[s]);
    }
    // level 1: histograms
#pragma unroll
    for (int set = 0; set < NSETS; ++set)
#pragma unroll
        for (int i = 0; i < kSamplePerThread; ++i) {
            const uint32_t k = keys[set][i];
            if (k != 0xFFFFFFFFu) atomicAdd(&sh->two.hist[set][bin_of(k, origin[set], scale[set])], 1u);
        }
    __syncthreads();
    const int wave = threadIdx.x / kWave;
    if (wave < 4 && want[wave < 4 ? wave : 0]) {
        const int set = (wave >> 1) ? NSETS - 1 : 0;
        uint32_t b;
        unsigned long long rb;
        scan_pick(sh->two.hist[set], rank[wave], b, rb);
        if (lane_id() == 0) {
            sh->bin[wave] = b;
            sh->rank_in_bin[wave] = rb;
        }
    }
    __syncthreads();
    // level 2: gather the picked bins, rank-count inside them
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!want[j]) continue;
        const int set = (j >> 1) ? NSETS - 1 : 0;
        const uint32_t b = sh->bin[j];
#pragma unroll
        for (int i = 0; i < kSamplePerThread; ++i) {
            const uint32_t k = keys[set][i];
            if (k != 0xFFFFFFFFu && bin_of(k, origin[set], scale[set]) == b) {
                const uint32_t idx = atomicAdd(&sh->count[j], 1u);
                if (idx < (uint32_t)kShortList) sh->two.list[j][idx] = k;
            }
        }
    }
    __syncthreads();
    bool crowded = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!want[j]) continue;
        if (sh->count[j] > (uint32_t)kShortList) crowded = true;
        else rank_pick(sh->two.list[j], sh->count[j], sh->rank_in_bin[j], &sh->result[j]);
    }
    __syncthreads();
    uint32_t res[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) res[j] = sh->result[j];
    if (crowded) {   // uniform: the counts live in LDS
        __syncthreads();
        SelectJob job[4];
        const int set_of[4] = {0, 0, NSETS - 1, NSETS - 1};
        int n_mine[NSETS];
#pragma unroll
        for (int set = 0; set < NSETS; ++set) n_mine[set] = kSamplePerThread;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int set = (j >> 1) ? NSETS - 1 : 0;
            job[j] = SelectJob{want[j] ? (uint32_t)kSample : 0u, rank[j], set_lo[set], set_hi[set]};
        }
        multi_select<4, NSETS, kSamplePerThread>(keys, n_mine, set_of, job, res, &sh->sel);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int set = s ? NSETS - 1 : 0;
        lo[s] = want[2 * s] ? res[2 * s] : 0u;
        hi[s] = want[2 * s + 1] ? res[2 * s + 1] : 0xFFFFFFFFu;
        // candidate bins span the bracket; an open side is bounded by the sample's extreme (the few keys beyond it
        // fall into the end bin)
        const uint32_t range_lo = want[2 * s] ? res[2 * s] : set_lo[set], range_hi = want[2 * s + 1] ? res[2 * s + 1] : set_hi[set];
        bin_origin[s] = bin_origin_for(range_lo);
        bin_scale[s] = m_valid > 0 ? bin_scale_for(range_lo, range_hi) : 0.0;
    }
}

// Raw moments -> unbiased covariance (torch_backend.py:395-397) -> plane vectors, columns [1,2] of eigh
// (torch_backend.py:415), sign convention: positive component sum.  mom[0..9] masked set, mom[10..19] all pixels;
// fewer than 3 masked pixels -> all pixels when allow_fallback (torch_backend.py:409-410).
__device__ void plane_from_moments(const double* mom, bool allow_fallback, double cov[9], float vecs[6], bool& use_all, unsigned long long& n_sel) {
    use_all = allow_fallback && mom[0] < 3.0;
    const double* a = use_all ? mom + 10 : mom;
    const double cnt = a[0];
    if (cnt > 1.0) {
        const double m0 = a[1] / cnt, m1 = a[2] / cnt, m2 = a[3] / cnt, d = cnt - 1.0;
        cov[0] = (a[4] - a[1] * m0) / d;
        cov[1] = cov[3] = (a[5] - a[1] * m1) / d;
        cov[2] = cov[6] = (a[6] - a[1] * m2) / d;
        cov[4] = (a[7] - a[2] * m1) / d;
        cov[5] = cov[7] = (a[8] - a[2] * m2) / d;
        cov[8] = (a[9] - a[3] * m2) / d;
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) cov[i] = 0.0;
    }
    double w[3], q[9];
    jacobi_eigh3(cov, w, q);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int src = c + 1;
        const double sum = q[0 * 3 + src] + q[1 * 3 + src] + q[2 * 3 + src];
        const double sgn = sum < 0.0 ? -1.0 : 1.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) vecs[r * 2 + c] = (float)(sgn * q[r * 3 + src]);
    }
    n_sel = (unsigned long long)cnt;
}

// ------------------------------------------------------------------------------------------------
// per-tile kernel A: moments -> covariance -> plane vectors; angle brackets from the sample
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kGroupThreads) void plane_kernel(const T* __restrict__ images, Geometry g, Workspace ws, int allow_fallback) {
    const int group = blockIdx.x;
    GroupState& st = ws.state[group];
    const GroupPixels gp = group_pixels(g, group);
    __shared__ double mom[kMoments];
    __shared__ SampleShared sample_sh;
    __shared__ float v_s[6];
    __shared__ int use_all_s;
    __shared__ unsigned long long n_sel_s;

    // strided sample of the group's pixels: issue the scattered loads first, they land while the moments are
    // summed and the covariance is diagonalised
    const int m = (int)min((int64_t)kSample, gp.count);
    float sample[kSamplePerThread][3];
#pragma unroll
    for (int i = 0; i < kSamplePerThread; ++i) {
        const int j = threadIdx.x + i * kGroupThreads;
        sample[i][0] = sample[i][1] = sample[i][2] = 0.0f;
        if (j < m) {
            int64_t tile, p;
            gp.locate(sample_position(j, m, gp.count), tile, p);
            load_od_scalar<T>(images, g.pixels, tile, p, sample[i]);
        }
    }
    SX_STAMP(st, 0);
    {
        // fixed-order (deterministic) sum of the workgroup partials: lanes fetch them in parallel, one thread
        // per moment adds them in index order
        __shared__ double stage[32][kMoments];
        const int64_t first = g.pooled ? 0 : (int64_t)group * g.blocks_per_tile;
        const int64_t nblk = g.pooled ? g.n_tiles * g.blocks_per_tile : g.blocks_per_tile;
        double running = 0.0;
        for (int64_t b0 = 0; b0 < nblk; b0 += 32) {
            const int live = (int)min((int64_t)32, nblk - b0);
            if ((int)threadIdx.x < live * kMoments) stage[threadIdx.x / kMoments][threadIdx.x % kMoments] = ws.partial[(first + b0) * kMoments + threadIdx.x];
            __syncthreads();
            if (threadIdx.x < kMoments)
                for (int b = 0; b < live; ++b) running += stage[b][threadIdx.x];
            __syncthreads();
        }
        if (threadIdx.x < kMoments) mom[threadIdx.x] = running;
    }
    __syncthreads();

    SX_STAMP(st, 1);
    if (threadIdx.x == 0) {
        double cov[9];
        bool use_all;
        unsigned long long n_sel_local;
        plane_from_moments(mom, allow_fallback != 0, cov, v_s, use_all, n_sel_local);
        const double cnt = (double)n_sel_local;
        use_all_s = use_all ? 1 : 0;
        n_sel_s = (unsigned long long)cnt;
#pragma unroll
        for (int k = 0; k < kMoments; ++k) st.mom[k] = mom[k];
#pragma unroll
        for (int i = 0; i < 9; ++i) st.cov[i] = cov[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) st.vecs[i] = v_s[i];
        st.use_all = use_all_s;
        st.n_sel = n_sel_s;
        st.fell_back = 0;
        st.hist_bad = 0;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) st.below[s] = st.ncand[s] = 0;
    }
    __syncthreads();

    SX_STAMP(st, 2);
    // angle keys of the selected sample pixels; the sample's OD is kept for the concentration brackets
    const bool use_all = use_all_s != 0;
    float v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = v_s[i];
    uint32_t keys[1][kSamplePerThread];
    float* sample_out = ws.sample_od + (size_t)group * 3 * kSample;
#pragma unroll
    for (int i = 0; i < kSamplePerThread; ++i) {
        const int j = threadIdx.x + i * kGroupThreads;
        keys[0][i] = (j < m && od_selected(sample[i], use_all)) ? angle_key(sample[i], v) : 0xFFFFFFFFu;
#pragma unroll
        for (int c = 0; c < 3; ++c) sample_out[c * kSample + j] = sample[i][c];
    }
    const unsigned long long n_sel = n_sel_s;
    const unsigned long long k0[2] = {nearest_rank_index(1.0, n_sel), nearest_rank_index(99.0, n_sel)};   // alpha = 1 (torch_backend.py:421-422)
    uint32_t lo[2], hi[2];
    double b_origin[2], b_scale[2];
    SX_STAMP(st, 3);
    sample_brackets<1>(keys, n_sel, k0, lo, hi, b_origin, b_scale, &sample_sh);
    SX_STAMP(st, 4);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            st.rank[s] = k0[s];
            st.lo_key[s] = lo[s];
            st.hi_key[s] = hi[s];
            st.bin_origin[s] = b_origin[s];
            st.bin_scale[s] = b_scale[s];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// streaming pass: count keys below each bracket, gather the keys inside it
//   kConc == false: slots 0,1 share the angle key of the selected pixels
//   kConc == true : slots 2,3 use the two concentrations of every pixel
// Candidates are first queued in LDS; one global atomic per workgroup and slot reserves their place.
// ------------------------------------------------------------------------------------------------
constexpr int kLocalCap = 4096;        // LDS queue per slot = pixels a workgroup handles between two flushes

struct LocalQueue {
    uint32_t count[2];
    uint32_t base[2];
    uint32_t keys[2][kLocalCap];
    uint32_t hist[2][256];        // bracket-relative histogram of everything this workgroup queued
};

template <typename T, int V, bool kConc>
__global__ __launch_bounds__(kStreamThreads) void bracket_kernel(const T* __restrict__ images, Geometry g, Workspace ws) {
    const int64_t tile = blockIdx.x / g.blocks_per_tile;
    const int chunk_id = blockIdx.x % g.blocks_per_tile;
    const int group = g.pooled ? 0 : (int)tile;
    GroupState& st = ws.state[group];
    constexpr int s0 = kConc ? 2 : 0;
    const int64_t p_begin = (int64_t)chunk_id * g.chunk;
    const int64_t p_end = min(p_begin + g.chunk, g.pixels);
    const T* img = images + tile * 3 * g.pixels;

    __shared__ LocalQueue queue;
    __shared__ uint32_t red[2][kStreamThreads / kWave];
    if (threadIdx.x < 2) queue.count[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < 512; i += kStreamThreads) (&queue.hist[0][0])[i] = 0;
    __syncthreads();

    float coef[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) coef[i] = kConc ? st.pinv[i] : st.vecs[i];
    const bool use_all = kConc ? true : (st.use_all != 0);
    const uint32_t lo_a = st.lo_key[s0], hi_a = st.hi_key[s0], lo_b = st.lo_key[s0 + 1], hi_b = st.hi_key[s0 + 1];
    const double scale_a = st.bin_scale[s0], scale_b = st.bin_scale[s0 + 1], origin_a = st.bin_origin[s0], origin_b = st.bin_origin[s0 + 1];
    uint32_t* cand_a = ws.cand + ((size_t)group * kSlots + s0) * kCap;
    uint32_t* cand_b = cand_a + kCap;
    uint32_t below_a = 0, below_b = 0;

    // The chunk is walked in phases of kLocalCap pixels; after each phase the LDS queues (which therefore
    // cannot overflow) are moved to the tile's candidate buffers with ONE global atomic per slot.
    constexpr int kPhasePixels = kLocalCap;
    for (int64_t phase_begin = p_begin; phase_begin < p_end; phase_begin += kPhasePixels) {
        const int64_t phase_end = min(phase_begin + kPhasePixels, p_end);
        for (int64_t p = phase_begin + (int64_t)threadIdx.x * V; p < phase_end; p += (int64_t)kStreamThreads * V) {
            float u[3][V];
#pragma unroll
            for (int c = 0; c < 3; ++c) load_unit<T, V>(img + c * g.pixels + p, u[c]);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float od[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) od[c] = optical_density(u[c][i]);
                const bool valid = od_selected(od, use_all);
                uint32_t key_a, key_b;
                if constexpr (kConc) {
                    float c0, c1;
                    concentration(od, coef, c0, c1);
                    key_a = float_key(c0);
                    key_b = float_key(c1);
                } else {
                    key_a = key_b = angle_key(od, coef);
                }
                below_a += (valid && key_a < lo_a) ? 1u : 0u;
                below_b += (valid && key_b < lo_b) ? 1u : 0u;
                if (valid && key_a >= lo_a && key_a <= hi_a) queue.keys[0][atomicAdd(&queue.count[0], 1u)] = key_a;
                if (valid && key_b >= lo_b && key_b <= hi_b) queue.keys[1][atomicAdd(&queue.count[1], 1u)] = key_b;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            const uint32_t n_local = queue.count[threadIdx.x];
            queue.base[threadIdx.x] = n_local ? atomicAdd(&st.ncand[s0 + threadIdx.x], n_local) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            uint32_t* dst = which == 0 ? cand_a : cand_b;
            const uint32_t n_local = queue.count[which], base = queue.base[which];
            const double origin = which == 0 ? origin_a : origin_b, scale = which == 0 ? scale_a : scale_b;
            for (uint32_t i = threadIdx.x; i < n_local; i += kStreamThreads) {
                const uint32_t key = queue.keys[which][i];
                if (base + i < (uint32_t)kCap) dst[base + i] = key;
                atomicAdd(&queue.hist[which][bin_of(key, origin, scale)], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 2) queue.count[threadIdx.x] = 0;
        __syncthreads();
    }

    const uint32_t wa = wave_sum_u32(below_a), wb = wave_sum_u32(below_b);
    if (lane_id() == 0) {
        red[0][threadIdx.x / kWave] = wa;
        red[1][threadIdx.x / kWave] = wb;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t sum = 0;
#pragma unroll
        for (int w = 0; w < kStreamThreads / kWave; ++w) sum += red[threadIdx.x][w];
        if (sum) atomicAdd(&st.below[s0 + threadIdx.x], sum);
    }
    // the workgroup's histograms are stored -- not added -- so the per-tile kernel can sum them in a fixed
    // order without any global atomic
    uint32_t* hist_out = ws.block_hist + (size_t)blockIdx.x * 512;
    for (int i = threadIdx.x; i < 512; i += kStreamThreads) hist_out[i] = (&queue.hist[0][0])[i];
}

// ------------------------------------------------------------------------------------------------
// exact order statistics of two slots: from the gathered candidates when both brackets held, else the
// failing slot radix-selects over the whole group (slow, exact)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ uint32_t select_whole_group(const T* __restrict__ images, const Geometry& g, const GroupState& st, int group, int slot, SelectShared<1>* sh) {
    const GroupPixels gp = group_pixels(g, group);
    float coef[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) coef[i] = slot < 2 ? st.vecs[i] : st.pinv[i];
    const bool use_all = slot < 2 ? (st.use_all != 0) : true;
    return radix_select_stream((unsigned long long)gp.count, st.rank[slot],
                               [&](unsigned long long i, uint32_t& k) {
                                   int64_t tile, p;
                                   gp.locate((int64_t)i, tile, p);
                                   float od[3];
                                   load_od_scalar<T>(images, g.pixels, tile, p, od);
                                   if (!od_selected(od, use_all)) return false;
                                   if (slot < 2) {
                                       k = angle_key(od, coef);
                                   } else {
                                       float c0, c1;
                                       concentration(od, coef, c0, c1);
                                       k = float_key(slot == 2 ? c0 : c1);
                                   }
                                   return true;
                               },
                               sh);
}

constexpr int kCandPerThread = kCap / kGroupThreads;   // 16 candidate keys per thread and slot, in registers

template <typename T>
__device__ void resolve_pair_radix(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int group, int first_slot, uint32_t (&key_out)[2], SelectShared<4>* sh) {
    GroupState& st = ws.state[group];
    SelectJob job[2];
    bool ok[2];
    uint32_t keys[2][kCandPerThread];
    int n_mine[2];
    const int set_of[2] = {0, 1};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int slot = first_slot + j;
        const uint32_t ncand = st.ncand[slot];
        const unsigned long long below = st.below[slot], want = st.rank[slot];
        ok[j] = ncand <= (uint32_t)kCap && want >= below && want - below < ncand;
        job[j] = SelectJob{ok[j] ? ncand : 0u, ok[j] ? want - below : 0ull, st.lo_key[slot], st.hi_key[slot]};
        const uint32_t* cand = ws.cand + ((size_t)group * kSlots + slot) * kCap;
        const uint32_t usable = ok[j] ? ncand : 0u;
        n_mine[j] = usable > threadIdx.x ? (int)((usable - threadIdx.x + kGroupThreads - 1) / kGroupThreads) : 0;
#pragma unroll
        for (int i = 0; i < kCandPerThread; ++i) {   // all loads issued before the first use
            const uint32_t idx = threadIdx.x + i * kGroupThreads;
            keys[j][i] = idx < usable ? cand[idx] : 0xFFFFFFFFu;
        }
    }
    multi_select<2, 2, kCandPerThread>(keys, n_mine, set_of, job, key_out, reinterpret_cast<SelectShared<2>*>(sh));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (ok[j]) continue;      // uniform across the workgroup
        __syncthreads();
        if (threadIdx.x == 0) atomicOr(&st.fell_back, 1u << (first_slot + j));
        key_out[j] = select_whole_group<T>(images, g, st, group, first_slot + j, reinterpret_cast<SelectShared<1>*>(sh));
    }
}

struct ResolveShared {
    union {
        SelectShared<4> sel;                  // radix fallback / whole-group select
        struct {
            uint32_t hist[2][256];
            uint32_t list[2][kShortList];
        } two;
    };
    uint32_t bin[2], count[2], result[2];
    unsigned long long rank_in_bin[2];
};

// Exact order statistics of two slots from the candidates gathered by the streaming pass, in two levels:
// sum the workgroups' bracket-relative histograms (fixed order), one wave per slot picks the bin holding the
// wanted rank, the candidates of that bin (~n/256 keys) are gathered and rank-counted.  Anything unusual --
// bracket missed or overflowed, a spilled queue, a crowded bin -- goes to the radix / whole-group paths.
template <typename T>
__device__ void resolve_pair(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int group, int first_slot, uint32_t (&key_out)[2], ResolveShared* sh) {
    GroupState& st = ws.state[group];
    bool ok[2];
    uint32_t ncand[2];
    double origin[2], scale[2];
    unsigned long long want_in[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int slot = first_slot + j;
        ncand[j] = st.ncand[slot];
        const unsigned long long below = st.below[slot], want = st.rank[slot];
        ok[j] = ncand[j] <= (uint32_t)kCap && want >= below && want - below < ncand[j] && ((st.hist_bad >> slot) & 1u) == 0;
        want_in[j] = ok[j] ? want - below : 0ull;
        origin[j] = st.bin_origin[slot];
        scale[j] = st.bin_scale[slot];
    }
    bool simple = ok[0] && ok[1];
    if (simple) {
        const int64_t first = g.pooled ? 0 : (int64_t)group * g.blocks_per_tile;
        const int64_t nblk = g.pooled ? g.n_tiles * g.blocks_per_tile : g.blocks_per_tile;
        if (threadIdx.x < 2) sh->count[threadIdx.x] = 0;
        if (threadIdx.x < 512) {                       // thread t owns bin t%256 of slot t/256
            const uint32_t* src = ws.block_hist + first * 512 + threadIdx.x;
            uint32_t sum = 0;
#pragma unroll 8
            for (int64_t b = 0; b < nblk; ++b) sum += src[b * 512];
            (&sh->two.hist[0][0])[threadIdx.x] = sum;
        }
        __syncthreads();
        if (first_slot == 2) SX_STAMP(st, 5);
        const int wave = threadIdx.x / kWave;
        if (wave < 2) {
            uint32_t b;
            unsigned long long rb;
            scan_pick(sh->two.hist[wave], want_in[wave], b, rb);
            if (lane_id() == 0) {
                sh->bin[wave] = b;
                sh->rank_in_bin[wave] = rb;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t* cand = ws.cand + ((size_t)group * kSlots + first_slot + j) * kCap;
            const uint32_t b = sh->bin[j];
            for (uint32_t base = threadIdx.x; base < ncand[j]; base += kGroupThreads * 8) {
                uint32_t k[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {          // eight independent loads in flight
                    const uint32_t idx = base + u * kGroupThreads;
                    k[u] = idx < ncand[j] ? cand[idx] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t idx = base + u * kGroupThreads;
                    if (idx < ncand[j] && bin_of(k[u], origin[j], scale[j]) == b) {
                        const uint32_t at = atomicAdd(&sh->count[j], 1u);
                        if (at < (uint32_t)kShortList) sh->two.list[j][at] = k[u];
                    }
                }
            }
        }
        __syncthreads();
        if (first_slot == 2) SX_STAMP(st, 11);
        if (sh->count[0] > (uint32_t)kShortList || sh->count[1] > (uint32_t)kShortList) {
            simple = false;                            // crowded bin (heavy ties): radix rounds instead
        } else {
            rank_pick(sh->two.list[0], sh->count[0], sh->rank_in_bin[0], &sh->result[0]);
            rank_pick(sh->two.list[1], sh->count[1], sh->rank_in_bin[1], &sh->result[1]);
            __syncthreads();
            if (first_slot == 2) SX_STAMP(st, 14);
            key_out[0] = sh->result[0];
            key_out[1] = sh->result[1];
        }
    }
    if (!simple) {
        __syncthreads();
        resolve_pair_radix<T>(images, g, ws, group, first_slot, key_out, &sh->sel);
    }
}

// ------------------------------------------------------------------------------------------------
// per-tile kernel B: angle percentiles -> HE_source -> pseudo-inverse; concentration brackets
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kGroupThreads) void stain_kernel(const T* __restrict__ images, Geometry g, Workspace ws) {
    const int group = blockIdx.x;
    GroupState& st = ws.state[group];
    const GroupPixels gp = group_pixels(g, group);
    __shared__ union {
        SampleShared sample;
        ResolveShared resolve;
    } lds;
    SampleShared& sample_sh = lds.sample;
    __shared__ float pinv_s[6];

    uint32_t phi_key[2];
    SX_STAMP(st, 6);
    resolve_pair<T>(images, g, ws, group, 0, phi_key, &lds.resolve);
    __syncthreads();
    SX_STAMP(st, 7);

    if (threadIdx.x == 0) {
        const float phi_lo = key_float(phi_key[0]), phi_hi = key_float(phi_key[1]);
        float he[6];
        stain_vectors_and_pinv(st.vecs, phi_lo, phi_hi, he, pinv_s);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            st.he[i] = he[i];
            st.pinv[i] = pinv_s[i];
        }
        st.phi[0] = phi_lo;
        st.phi[1] = phi_hi;
        st.ncand_seen[0] = st.ncand[0];
        st.ncand_seen[1] = st.ncand[1];
    }
    __syncthreads();

    SX_STAMP(st, 8);
    // concentration brackets from the same strided sample (every pixel takes part: torch_backend.py:442-448)
    const int m = (int)min((int64_t)kSample, gp.count);
    float pinv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) pinv[i] = pinv_s[i];
    uint32_t keys[2][kSamplePerThread];
    const float* sample_in = ws.sample_od + (size_t)group * 3 * kSample;
#pragma unroll
    for (int i = 0; i < kSamplePerThread; ++i) {
        const int j = threadIdx.x + i * kGroupThreads;
        uint32_t ka = 0xFFFFFFFFu, kb = 0xFFFFFFFFu;
        if (j < m) {
            const float od[3] = {sample_in[j], sample_in[kSample + j], sample_in[2 * kSample + j]};
            float c0, c1;
            concentration(od, pinv, c0, c1);
            ka = float_key(c0);
            kb = float_key(c1);
        }
        keys[0][i] = ka;
        keys[1][i] = kb;
    }
    const unsigned long long n_all = (unsigned long long)gp.count;
    const unsigned long long k99 = nearest_rank_index(99.0, n_all);          // torch_backend.py:447-448
    const unsigned long long k0[2] = {k99, k99};
    uint32_t lo[2], hi[2];
    double b_origin[2], b_scale[2];
    SX_STAMP(st, 9);
    sample_brackets<2>(keys, n_all, k0, lo, hi, b_origin, b_scale, &sample_sh);
    SX_STAMP(st, 10);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            st.rank[2 + s] = k99;
            st.lo_key[2 + s] = lo[s];
            st.hi_key[2 + s] = hi[s];
            st.bin_origin[2 + s] = b_origin[s];
            st.bin_scale[2 + s] = b_scale[s];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-tile kernel C: concentration percentiles -> scale factors (transform) / outputs (fit)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kGroupThreads) void scale_kernel(const T* __restrict__ images, Geometry g, Workspace ws, const float* __restrict__ target_max_conc, float* __restrict__ he_out, float* __restrict__ max_c_out) {
    const int group = blockIdx.x;
    GroupState& st = ws.state[group];
    __shared__ ResolveShared sel;
    uint32_t c_key[2];
    SX_STAMP(st, 12);
    resolve_pair<T>(images, g, ws, group, 2, c_key, &sel);
    SX_STAMP(st, 13);
    if (threadIdx.x == 0) {
        const float m0 = key_float(c_key[0]), m1 = key_float(c_key[1]);
        st.max_c[0] = m0;
        st.max_c[1] = m1;
        st.ncand_seen[2] = st.ncand[2];
        st.ncand_seen[3] = st.ncand[3];
        if (target_max_conc) {
            st.scale[0] = target_max_conc[0] / m0;      // torch_backend.py:452
            st.scale[1] = target_max_conc[1] / m1;
        }
        if (he_out) {
#pragma unroll
            for (int i = 0; i < 6; ++i) he_out[i] = st.he[i];
            max_c_out[0] = m0;
            max_c_out[1] = m1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// final pass: concentrations -> rescale -> reconstruct -> clamp -> cast  (torch_backend.py:452-461,560)
// ------------------------------------------------------------------------------------------------
template <typename T, typename O, int V, bool kUnit>
__global__ __launch_bounds__(kStreamThreads) void reconstruct_kernel(const T* __restrict__ images, O* __restrict__ out, Geometry g, Workspace ws, const float* __restrict__ stain_matrix) {
    const int64_t tile = blockIdx.x / g.blocks_per_tile;
    const int chunk_id = blockIdx.x % g.blocks_per_tile;
    const GroupState& st = ws.state[tile];
    const int64_t p_begin = (int64_t)chunk_id * g.chunk;
    const int64_t p_end = min(p_begin + g.chunk, g.pixels);
    const T* img = images + tile * 3 * g.pixels;
    O* dst = out + tile * 3 * g.pixels;

    float pinv[6], sm[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        pinv[i] = st.pinv[i];
        sm[i] = stain_matrix[i];
    }
    const float s0 = st.scale[0], s1 = st.scale[1];

    for (int64_t p = p_begin + (int64_t)threadIdx.x * V; p < p_end; p += (int64_t)kStreamThreads * V) {
        float u[3][V];
#pragma unroll
        for (int c = 0; c < 3; ++c) load_unit<T, V>(img + c * g.pixels + p, u[c]);
        O res[3][V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float od[3], c0, c1;
#pragma unroll
            for (int c = 0; c < 3; ++c) od[c] = optical_density(u[c][i]);
            concentration(od, pinv, c0, c1);
            c0 *= s0;                                                   // :453
            c1 *= s1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float od_new = fmaf(sm[c * 2 + 1], c1, sm[c * 2] * c0);         // :455
                float rgb = kIo * exp2f(-od_new * kLog2e);                            // :458  (v_exp_f32)
                rgb = fminf(fmaxf(rgb, 0.0f), 255.0f);                                // :459, :128
                if constexpr (kUnit) {
                    // cast to the input dtype first, then /255 in that dtype (_template.py:111-112);
                    // u8 promotes to f32
                    if constexpr (sizeof(T) == 1) {
                        res[c][i] = (float)Elem<T>::store(rgb) / 255.0f;
                    } else if constexpr (sizeof(T) == 8) {
                        res[c][i] = (double)rgb / 255.0;
                    } else {
                        res[c][i] = Elem<O>::store(Elem<T>::load(Elem<T>::store(rgb)) / 255.0f);
                    }
                } else {
                    res[c][i] = Elem<O>::store(rgb);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) store_pack<O, V>(dst + c * g.pixels + p, res[c]);
    }
}

// ------------------------------------------------------------------------------------------------
// distributed pooled fit: the batch is sharded over ranks, every reduction stage is one
// small all-reduce done by the host between these kernels.  Order statistics use a plain 4-round byte
// radix select whose 256-bin histograms are integer sums over ranks => the same (HE, maxC) bits on
// every rank, whatever the sharding.
// ------------------------------------------------------------------------------------------------
struct alignas(256) DFitState {
    double mom[kMoments];
    float vecs[6], he[6], pinv[6];
    float phi[2], max_c[2];
    unsigned long long n_sel, n_all;
    unsigned long long rank[kSlots];
    uint32_t prefix[kSlots], mask[kSlots];
    int round[2];
};

__global__ void dfit_reduce_partials_kernel(const double* __restrict__ partial, int64_t nblk, double* __restrict__ moments) {
    const int k = threadIdx.x;
    if (k >= kMoments) return;
    double s = 0.0;
    for (int64_t b = 0; b < nblk; ++b) s += partial[b * kMoments + k];
    moments[k] = s;
}

__global__ void dfit_begin_kernel(const double* __restrict__ moments, DFitState* __restrict__ st) {
    if (threadIdx.x != 0) return;
    for (int k = 0; k < kMoments; ++k) st->mom[k] = moments[k];
    double cov[9];
    bool use_all;
    unsigned long long n_sel;
    plane_from_moments(st->mom, false, cov, st->vecs, use_all, n_sel);
    st->n_sel = n_sel;
    st->n_all = (unsigned long long)st->mom[10];
    st->rank[0] = nearest_rank_index(1.0, n_sel);
    st->rank[1] = nearest_rank_index(99.0, n_sel);
    for (int s = 0; s < kSlots; ++s) st->prefix[s] = st->mask[s] = 0;
    st->round[0] = st->round[1] = 0;
}

template <typename T, int V>
__global__ __launch_bounds__(kStreamThreads) void dfit_histogram_kernel(const T* __restrict__ images, Geometry g, const DFitState* __restrict__ st, int stage, unsigned long long* __restrict__ hist) {
    const int64_t tile = blockIdx.x / g.blocks_per_tile;
    const int chunk_id = blockIdx.x % g.blocks_per_tile;
    const int64_t p_begin = (int64_t)chunk_id * g.chunk, p_end = min(p_begin + g.chunk, g.pixels);
    const T* img = images + tile * 3 * g.pixels;
    __shared__ uint32_t local[2][256];
    for (int i = threadIdx.x; i < 512; i += kStreamThreads) (&local[0][0])[i] = 0;
    __syncthreads();
    const int s0 = stage * 2;
    float coef[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) coef[i] = stage ? st->pinv[i] : st->vecs[i];
    const int shift = 24 - 8 * st->round[stage];
    const uint32_t pa = st->prefix[s0], ma = st->mask[s0], pb = st->prefix[s0 + 1], mb = st->mask[s0 + 1];
    for (int64_t p = p_begin + (int64_t)threadIdx.x * V; p < p_end; p += (int64_t)kStreamThreads * V) {
        float u[3][V];
#pragma unroll
        for (int c = 0; c < 3; ++c) load_unit<T, V>(img + c * g.pixels + p, u[c]);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float od[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) od[c] = optical_density(u[c][i]);
            uint32_t ka, kb;
            if (stage) {
                float c0, c1;
                concentration(od, coef, c0, c1);
                ka = float_key(c0);
                kb = float_key(c1);
            } else {
                if (!od_selected(od, false)) continue;
                ka = kb = angle_key(od, coef);
            }
            if (((ka ^ pa) & ma) == 0) atomicAdd(&local[0][(ka >> shift) & 255u], 1u);
            if (((kb ^ pb) & mb) == 0) atomicAdd(&local[1][(kb >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += kStreamThreads) {
        const uint32_t v = (&local[0][0])[i];
        if (v) atomicAdd(&hist[i], (unsigned long long)v);
    }
}

__global__ __launch_bounds__(128) void dfit_advance_kernel(DFitState* __restrict__ st, int stage, const unsigned long long* __restrict__ hist) {
    __shared__ uint32_t bins[2][256];
    __shared__ unsigned long long carry[2][256];
    const int s0 = stage * 2;
    // histograms can exceed 2^32 per bin in principle: scan in 64 bit with one thread per slot
    if (threadIdx.x < 2) {
        const int j = threadIdx.x;
        unsigned long long r = st->rank[s0 + j], cum = 0;
        int d = 0;
        for (; d < 255; ++d) {
            const unsigned long long h = hist[j * 256 + d];
            if (cum + h > r) break;
            cum += h;
        }
        const int shift = 24 - 8 * st->round[stage];
        st->prefix[s0 + j] |= (uint32_t)d << shift;
        st->mask[s0 + j] |= 0xFFu << shift;
        st->rank[s0 + j] = r - cum;
    }
    (void)bins;
    (void)carry;
    __syncthreads();
    if (threadIdx.x != 0) return;
    st->round[stage] += 1;
    if (st->round[stage] < 4) return;
    if (stage == 0) {
        const float phi_lo = key_float(st->prefix[0]), phi_hi = key_float(st->prefix[1]);
        st->phi[0] = phi_lo;
        st->phi[1] = phi_hi;
        stain_vectors_and_pinv(st->vecs, phi_lo, phi_hi, st->he, st->pinv);
        st->rank[2] = st->rank[3] = nearest_rank_index(99.0, st->n_all);
    } else {
        st->max_c[0] = key_float(st->prefix[2]);
        st->max_c[1] = key_float(st->prefix[3]);
    }
}

__global__ void dfit_result_kernel(const DFitState* __restrict__ st, float* __restrict__ he_out, float* __restrict__ max_c_out) {
    if (threadIdx.x < 6) he_out[threadIdx.x] = st->he[threadIdx.x];
    if (threadIdx.x < 2) max_c_out[threadIdx.x] = st->max_c[threadIdx.x];
}

__global__ void export_params_kernel(const GroupState* __restrict__ state, int64_t n_groups, float* __restrict__ out) {
    const int64_t gidx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (gidx >= n_groups) return;
    const GroupState& st = state[gidx];
    float* o = out + gidx * SX_MACENKO_PARAM_FLOATS;
    o[0] = (float)st.n_sel;
    o[1] = (float)st.use_all;
    for (int i = 0; i < 6; ++i) o[2 + i] = st.vecs[i];
    o[8] = st.phi[0];
    o[9] = st.phi[1];
    for (int i = 0; i < 6; ++i) o[10 + i] = st.he[i];
    o[16] = st.max_c[0];
    o[17] = st.max_c[1];
    o[18] = (float)st.fell_back;
    for (int s = 0; s < kSlots; ++s) o[19 + s] = (float)st.ncand_seen[s];
    for (int i = 0; i < 9; ++i) o[23 + i] = (float)st.cov[i];
    for (int i = 0; i < 16; ++i) o[32 + i] = (float)((double)(st.stamp[i] - st.stamp[0]) * 0.01);   // us (100 MHz clock)
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static bool aligned_for(const void* p, size_t bytes) { return (reinterpret_cast<uintptr_t>(p) % bytes) == 0; }

template <typename T, int V>
static int run_estimate(const T* images, const Geometry& g, const Workspace& ws, int n_groups, int allow_fallback, const float* tmc, float* he_out, float* max_c_out, hipStream_t stream) {
    const unsigned grid = (unsigned)(g.n_tiles * g.blocks_per_tile);
    hipLaunchKernelGGL((stats_kernel<T, V>), dim3(grid), dim3(kStreamThreads), 0, stream, images, g, ws.partial);
    hipLaunchKernelGGL((plane_kernel<T>), dim3(n_groups), dim3(kGroupThreads), 0, stream, images, g, ws, allow_fallback);
    hipLaunchKernelGGL((bracket_kernel<T, V, false>), dim3(grid), dim3(kStreamThreads), 0, stream, images, g, ws);
    hipLaunchKernelGGL((stain_kernel<T>), dim3(n_groups), dim3(kGroupThreads), 0, stream, images, g, ws);
    hipLaunchKernelGGL((bracket_kernel<T, V, true>), dim3(grid), dim3(kStreamThreads), 0, stream, images, g, ws);
    hipLaunchKernelGGL((scale_kernel<T>), dim3(n_groups), dim3(kGroupThreads), 0, stream, images, g, ws, tmc, he_out, max_c_out);
    return check_launch("macenko estimate");
}

template <typename T, typename O, int V>
static int run_transform(const T* images, O* out, const Geometry& g, const Workspace& ws, const float* sm, const float* tmc, bool unit, hipStream_t stream) {
    int rc = run_estimate<T, V>(images, g, ws, (int)g.n_tiles, 1, tmc, nullptr, nullptr, stream);
    if (rc != SX_OK) return rc;
    const unsigned grid = (unsigned)(g.n_tiles * g.blocks_per_tile);
    if (unit)
        hipLaunchKernelGGL((reconstruct_kernel<T, O, V, true>), dim3(grid), dim3(kStreamThreads), 0, stream, images, out, g, ws, sm);
    else
        hipLaunchKernelGGL((reconstruct_kernel<T, O, V, false>), dim3(grid), dim3(kStreamThreads), 0, stream, images, out, g, ws, sm);
    return check_launch("macenko reconstruct");
}

template <typename T>
static int transform_typed(const void* images, void* out, const Geometry& g0, const Workspace& ws, const float* sm, const float* tmc, bool unit, hipStream_t stream) {
    Geometry g = g0;
    const bool u8_unit = unit && sizeof(T) == 1;
    const size_t out_elem = u8_unit ? sizeof(float) : sizeof(T);
    const bool vec = (g.pixels % 4 == 0) && aligned_for(images, sizeof(T) * 4) && aligned_for(out, out_elem * 4);
    g.chunk = kStreamThreads * (vec ? 4 : 1) * kIters * (vec ? 1 : 4);   // same pixels per workgroup on both paths
    const T* in = static_cast<const T*>(images);
    if constexpr (sizeof(T) == 1) {
        if (u8_unit) {
            return vec ? run_transform<T, float, 4>(in, static_cast<float*>(out), g, ws, sm, tmc, true, stream)
                       : run_transform<T, float, 1>(in, static_cast<float*>(out), g, ws, sm, tmc, true, stream);
        }
    }
    return vec ? run_transform<T, T, 4>(in, static_cast<T*>(out), g, ws, sm, tmc, unit, stream)
               : run_transform<T, T, 1>(in, static_cast<T*>(out), g, ws, sm, tmc, unit, stream);
}

template <typename T>
static int fit_typed(const void* images, const Geometry& g0, const Workspace& ws, float* he_out, float* max_c_out, hipStream_t stream) {
    Geometry g = g0;
    const bool vec = (g.pixels % 4 == 0) && aligned_for(images, sizeof(T) * 4);
    g.chunk = kStreamThreads * (vec ? 4 : 1) * kIters * (vec ? 1 : 4);
    const T* in = static_cast<const T*>(images);
    return vec ? run_estimate<T, 4>(in, g, ws, 1, 0, nullptr, he_out, max_c_out, stream)
               : run_estimate<T, 1>(in, g, ws, 1, 0, nullptr, he_out, max_c_out, stream);
}


// ---- distributed pooled fit: staged entry points (host does the all-reduces in between) -----------
template <typename T>
static int dfit_moments_typed(const void* images, const Geometry& g0, const Workspace& ws, double* moments, hipStream_t stream) {
    Geometry g = g0;
    const bool vec = (g.pixels % 4 == 0) && aligned_for(images, sizeof(T) * 4);
    g.chunk = kStreamThreads * 4 * kIters;
    const unsigned grid = (unsigned)(g.n_tiles * g.blocks_per_tile);
    const T* in = static_cast<const T*>(images);
    if (vec)
        hipLaunchKernelGGL((stats_kernel<T, 4>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, ws.partial);
    else
        hipLaunchKernelGGL((stats_kernel<T, 1>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, ws.partial);
    hipLaunchKernelGGL(dfit_reduce_partials_kernel, dim3(1), dim3(64), 0, stream, ws.partial, (int64_t)grid, moments);
    return check_launch("macenko dfit moments");
}

template <typename T>
static int dfit_histogram_typed(const void* images, const Geometry& g0, const DFitState* st, int stage, unsigned long long* hist, hipStream_t stream) {
    Geometry g = g0;
    const bool vec = (g.pixels % 4 == 0) && aligned_for(images, sizeof(T) * 4);
    g.chunk = kStreamThreads * 4 * kIters;
    const unsigned grid = (unsigned)(g.n_tiles * g.blocks_per_tile);
    const T* in = static_cast<const T*>(images);
    if (hipMemsetAsync(hist, 0, 512 * sizeof(unsigned long long), stream) != hipSuccess) return fail(SX_ERR_LAUNCH, "hipMemsetAsync failed");
    if (vec)
        hipLaunchKernelGGL((dfit_histogram_kernel<T, 4>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, st, stage, hist);
    else
        hipLaunchKernelGGL((dfit_histogram_kernel<T, 1>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, st, stage, hist);
    return check_launch("macenko dfit histogram");
}

}  // namespace macenko
}  // namespace sx

using namespace sx;
using namespace sx::macenko;

static int validate_images(const void* images, int64_t n, int64_t h, int64_t w, const void* ws, size_t ws_bytes, size_t need) {
    if (!images) return fail(SX_ERR_BAD_ARG, "images pointer is null");
    if (n <= 0 || h <= 0 || w <= 0) return fail(SX_ERR_BAD_ARG, "images must be (N,3,H,W) with positive sizes, got N=%lld H=%lld W=%lld", (long long)n, (long long)h, (long long)w);
    if (n * h * w >= (1ll << 32)) return fail(SX_ERR_BAD_ARG, "N*H*W must be below 2^32 pixels");
    if (!ws || ws_bytes < need) return fail(SX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, ws_bytes);
    if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return fail(SX_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    return SX_OK;
}

extern "C" size_t sx_macenko_workspace_bytes(int64_t n_tiles, int64_t height, int64_t width) {
    if (n_tiles <= 0 || height <= 0 || width <= 0) return 0;
    return macenko::workspace_bytes(n_tiles, height * width);
}

extern "C" int sx_macenko_transform(const void* images, void* out, int dtype, int64_t n, int64_t h, int64_t w, const float* sm, const float* tmc, unsigned flags, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    int rc = validate_images(images, n, h, w, ws_ptr, ws_bytes, sx_macenko_workspace_bytes(n, h, w));
    if (rc != SX_OK) return rc;
    if (!out || !sm || !tmc) return fail(SX_ERR_BAD_ARG, "out / stain_matrix / target_max_conc pointer is null");
    Geometry g{n, h * w, blocks_per_tile_for(h * w), 0, 0};
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    const bool unit = (flags & SX_MACENKO_NORMALIZE_0_1) != 0;
    switch (dtype) {
        case SX_U8: return transform_typed<uint8_t>(images, out, g, ws, sm, tmc, unit, stream);
        case SX_F16: return transform_typed<__half>(images, out, g, ws, sm, tmc, unit, stream);
        case SX_BF16: return transform_typed<__hip_bfloat16>(images, out, g, ws, sm, tmc, unit, stream);
        case SX_F32: return transform_typed<float>(images, out, g, ws, sm, tmc, unit, stream);
        case SX_F64: return transform_typed<double>(images, out, g, ws, sm, tmc, unit, stream);
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
}

extern "C" int sx_macenko_fit(const void* images, int dtype, int64_t n, int64_t h, int64_t w, float* he_out, float* max_c_out, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    int rc = validate_images(images, n, h, w, ws_ptr, ws_bytes, sx_macenko_workspace_bytes(n, h, w));
    if (rc != SX_OK) return rc;
    if (!he_out || !max_c_out) return fail(SX_ERR_BAD_ARG, "he_out / max_c_out pointer is null");
    Geometry g{n, h * w, blocks_per_tile_for(h * w), 0, 1};
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    switch (dtype) {
        case SX_U8: return fit_typed<uint8_t>(images, g, ws, he_out, max_c_out, stream);
        case SX_F16: return fit_typed<__half>(images, g, ws, he_out, max_c_out, stream);
        case SX_BF16: return fit_typed<__hip_bfloat16>(images, g, ws, he_out, max_c_out, stream);
        case SX_F32: return fit_typed<float>(images, g, ws, he_out, max_c_out, stream);
        case SX_F64: return fit_typed<double>(images, g, ws, he_out, max_c_out, stream);
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
}

extern "C" int sx_macenko_tile_params(const void* ws_ptr, int64_t n_groups, float* params_out, void* stream_ptr) {
    if (!ws_ptr || !params_out || n_groups <= 0) return fail(SX_ERR_BAD_ARG, "bad argument to sx_macenko_tile_params");
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    const unsigned grid = (unsigned)((n_groups + 63) / 64);
    hipLaunchKernelGGL(export_params_kernel, dim3(grid), dim3(64), 0, stream, static_cast<const GroupState*>(ws_ptr), n_groups, params_out);
    return check_launch("macenko export_params");
}

extern "C" size_t sx_macenko_dfit_state_bytes(void) { return sizeof(DFitState); }

extern "C" int sx_macenko_dfit_moments(const void* images, int dtype, int64_t n, int64_t h, int64_t w, double* moments_out, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    int rc = validate_images(images, n, h, w, ws_ptr, ws_bytes, sx_macenko_workspace_bytes(n, h, w));
    if (rc != SX_OK) return rc;
    if (!moments_out) return fail(SX_ERR_BAD_ARG, "moments_out pointer is null");
    Geometry g{n, h * w, blocks_per_tile_for(h * w), 0, 1};
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    switch (dtype) {
        case SX_U8: return dfit_moments_typed<uint8_t>(images, g, ws, moments_out, stream);
        case SX_F16: return dfit_moments_typed<__half>(images, g, ws, moments_out, stream);
        case SX_BF16: return dfit_moments_typed<__hip_bfloat16>(images, g, ws, moments_out, stream);
        case SX_F32: return dfit_moments_typed<float>(images, g, ws, moments_out, stream);
        case SX_F64: return dfit_moments_typed<double>(images, g, ws, moments_out, stream);
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
}

extern "C" int sx_macenko_dfit_begin(const double* moments, void* state, void* stream_ptr) {
    if (!moments || !state) return fail(SX_ERR_BAD_ARG, "moments / state pointer is null");
    hipLaunchKernelGGL(dfit_begin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_ptr), moments, static_cast<DFitState*>(state));
    return check_launch("macenko dfit begin");
}

extern "C" int sx_macenko_dfit_histogram(const void* images, int dtype, int64_t n, int64_t h, int64_t w, const void* state, int stage, unsigned long long* hist_out, void* stream_ptr) {
    if (!images || !state || !hist_out) return fail(SX_ERR_BAD_ARG, "images / state / hist_out pointer is null");
    if (n <= 0 || h <= 0 || w <= 0 || (stage != 0 && stage != 1)) return fail(SX_ERR_BAD_ARG, "bad sizes or stage");
    Geometry g{n, h * w, blocks_per_tile_for(h * w), 0, 1};
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    const DFitState* st = static_cast<const DFitState*>(state);
    switch (dtype) {
        case SX_U8: return dfit_histogram_typed<uint8_t>(images, g, st, stage, hist_out, stream);
        case SX_F16: return dfit_histogram_typed<__half>(images, g, st, stage, hist_out, stream);
        case SX_BF16: return dfit_histogram_typed<__hip_bfloat16>(images, g, st, stage, hist_out, stream);
        case SX_F32: return dfit_histogram_typed<float>(images, g, st, stage, hist_out, stream);
        case SX_F64: return dfit_histogram_typed<double>(images, g, st, stage, hist_out, stream);
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
}

extern "C" int sx_macenko_dfit_advance(void* state, int stage, const unsigned long long* hist, void* stream_ptr) {
    if (!state || !hist || (stage != 0 && stage != 1)) return fail(SX_ERR_BAD_ARG, "bad argument to sx_macenko_dfit_advance");
    hipLaunchKernelGGL(dfit_advance_kernel, dim3(1), dim3(128), 0, static_cast<hipStream_t>(stream_ptr), static_cast<DFitState*>(state), stage, hist);
    return check_launch("macenko dfit advance");
}

extern "C" int sx_macenko_dfit_result(const void* state, float* he_out, float* max_c_out, void* stream_ptr) {
    if (!state || !he_out || !max_c_out) return fail(SX_ERR_BAD_ARG, "bad argument to sx_macenko_dfit_result");
    hipLaunchKernelGGL(dfit_result_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_ptr), static_cast<const DFitState*>(state), he_out, max_c_out);
    return check_launch("macenko dfit result");
}
