// tbk_fold.hip -- k lists with long runs of a shared component (uniform grids, planes, lines): fold the model
// along that component once per run and evaluate a model of lower dimension.
//
// The Fourier sum of /root/reference/src/tbmodels/_tb_model.py:1109-1122 costs 8 N^2 N_R flops per k-point
// whatever the k list looks like.  On a grid -- BASELINE config 4 is the 100 x 100 x 100 mesh of
// `np.meshgrid(..., indexing="ij")` -- 10^4 consecutive k-points share k_1, and for them
//
//     exp(2 pi i k.R) = exp(2 pi i k_1 R_1) * exp(2 pi i (k_2 R_2 + k_3 R_3))
//
// so the lattice vectors that differ only in R_1 can be summed ONCE per plane:
//
//     hop'[(R_2, R_3)] = sum_{R_1} exp(2 pi i k_1 R_1) hop[(R_1, R_2, R_3)]
//
// and the plane is a 2-D model with as many "lattice vectors" as there are distinct (R_2, R_3): 313 instead of
// 4096 half-space vectors for the synthetic headline model (|R_i| <= 12) -- 13x less work for the H(k) contraction,
// with every kernel of the path unchanged (phase rows, MFMA contraction, eigensolvers all run on the folded model).
//
// The fold acts on the STAGED operand Bt (two real rows P_r, Q_r per lattice vector: H = sum_r c_r P_r + s_r Q_r,
// tbk_stage.hip).  With theta = alpha + sigma beta' (alpha = 2 pi k_f R_f, beta' the phase of the canonical
// (dim-1)-vector rho', sigma = -1 where R's remaining components had to be negated to make them canonical):
//
//     P'_rho += cos(alpha) P_r + sin(alpha) Q_r          Q'_rho += sigma (-sin(alpha) P_r + cos(alpha) Q_r)
//
// One pass over Bt per run (272 MB at the headline shape: 55 us) against a contraction 13x shorter.

#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

#include "tbk_internal.h"

namespace {

// thread (column c of the flattened Bt row, folded vector rho): gathers the rows of its list
__global__ void __launch_bounds__(256)
fold_rows_kernel(const double* __restrict__ Bt, int64_t row_len, const int64_t* __restrict__ lptr,
                 const int32_t* __restrict__ lrec, const int32_t* __restrict__ rcomp, double k_f,
                 double* __restrict__ B2) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t rho = blockIdx.y;
    if (c >= row_len) return;
    double accp = 0.0, accq = 0.0;
    for (int64_t t = lptr[rho]; t < lptr[rho + 1]; ++t) {
        const int32_t rec = lrec[t];
        const int64_t r = rec & 0x7fffffff;
        const double sigma = rec < 0 ? -1.0 : 1.0;
        double sa, ca;
        sincospi(2.0 * k_f * (double)rcomp[r], &sa, &ca);  // uniform per (rho, t): exact argument reduction
        const double p = Bt[(2 * r) * row_len + c], q = Bt[(2 * r + 1) * row_len + c];
        accp = fma(ca, p, fma(sa, q, accp));
        accq = fma(sigma * ca, q, fma(-sigma * sa, p, accq));
    }
    B2[(2 * rho) * row_len + c] = accp;
    B2[(2 * rho + 1) * row_len + c] = accq;
}

constexpr int FOLD_GROUP = 16;

struct FoldValues {
    double v[FOLD_GROUP];  // shared-component value of every run of the group (a kernel argument: no copy to wait for)
};

// (cos, sin)(2 pi k_f[g] R_f[r]) for the runs of a group
__global__ void fold_table_kernel(const int32_t* __restrict__ rcomp, int64_t n_r, const FoldValues k_f, int n_g,
                                  double* __restrict__ table) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_r * n_g) return;
    const int64_t r = i / n_g;
    const int g = (int)(i % n_g);
    double sa, ca;
    sincospi(2.0 * k_f.v[g] * (double)rcomp[r], &sa, &ca);  // exact argument reduction
    table[2 * i] = ca;
    table[2 * i + 1] = sa;
}

// the same table for n_g lines whose shared-component values sit in device memory at k_f[g * stride]
__global__ void fold_table_dev_kernel(const int32_t* __restrict__ rcomp, int64_t n_r, const double* __restrict__ k_f,
                                      int64_t stride, int n_g, double* __restrict__ table) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_r * n_g) return;
    const int64_t r = i / n_g;
    const int g = (int)(i % n_g);
    double sa, ca;
    sincospi(2.0 * k_f[(int64_t)g * stride] * (double)rcomp[r], &sa, &ca);
    table[2 * i] = ca;
    table[2 * i + 1] = sa;
}

// The same fold for up to FOLD_GROUP runs in ONE pass over Bt: every (row pair, column) is loaded once and
// accumulated into the operand of each run of the group (its phases come from the table above).

__global__ void __launch_bounds__(256)
fold_rows_group_kernel(const double* __restrict__ Bt, int64_t row_len, const int64_t* __restrict__ lptr,
                       const int32_t* __restrict__ lrec, const double* __restrict__ table_all, int n_total,
                       int64_t b2_stride, double* __restrict__ B2_all) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t rho = blockIdx.y;
    if (c >= row_len) return;
    // blockIdx.z: which FOLD_GROUP slots of the n_total this block folds
    const int g_base = blockIdx.z * FOLD_GROUP;
    const int n_g = min(FOLD_GROUP, n_total - g_base);
    const double* table = table_all + 2 * g_base;
    double* B2 = B2_all + (size_t)g_base * b2_stride;
    double accp[FOLD_GROUP], accq[FOLD_GROUP];
#pragma unroll
    for (int g = 0; g < FOLD_GROUP; ++g) accp[g] = accq[g] = 0.0;
    for (int64_t t = lptr[rho]; t < lptr[rho + 1]; ++t) {
        const int32_t rec = lrec[t];
        const int64_t r = rec & 0x7fffffff;
        const double sigma = rec < 0 ? -1.0 : 1.0;
        const double p = Bt[(2 * r) * row_len + c], q = Bt[(2 * r + 1) * row_len + c];
        const double* tab = table + 2 * r * n_total;  // uniform
#pragma unroll
        for (int g = 0; g < FOLD_GROUP; ++g) {
            if (g < n_g) {
                const double ca = tab[2 * g], sa = tab[2 * g + 1];
                accp[g] = fma(ca, p, fma(sa, q, accp[g]));
                accq[g] = fma(sigma * ca, q, fma(-sigma * sa, p, accq[g]));
            }
        }
    }
#pragma unroll
    for (int g = 0; g < FOLD_GROUP; ++g) {
        if (g < n_g) {
            B2[g * b2_stride + (2 * rho) * row_len + c] = accp[g];
            B2[g * b2_stride + (2 * rho + 1) * row_len + c] = accq[g];
        }
    }
}

// k2[i][:] = k[i][all components but f]
__global__ void drop_component_kernel(const double* __restrict__ k, int dim, int f, int64_t nk, double* __restrict__ k2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nk) return;
    int o = 0;
    for (int d = 0; d < dim; ++d)
        if (d != f) k2[i * (dim - 1) + o++] = k[i * dim + d];
}

bool canonical_negate(std::vector<int32_t>& v) {
    for (int32_t x : v) {
        if (x > 0) return false;
        if (x < 0) {
            for (int32_t& y : v) y = -y;
            return true;
        }
    }
    return false;
}

}  // namespace

// Builds the folded lattice and the row lists for folding the lattice R[n_r][dim] along component f.
static int build_plan(tbk_fold_plan_t& plan, const int32_t* R, int64_t n_r, int dim, int f, int ncol_pad, int capacity) {
    if (plan.built) return TBK_OK;
    std::map<std::vector<int32_t>, int32_t> index;
    std::vector<std::vector<int32_t>> lists;
    std::vector<int32_t> rcomp((size_t)n_r);
    plan.h_R2.clear();
    for (int64_t r = 0; r < n_r; ++r) {
        std::vector<int32_t> rho;
        for (int d = 0; d < dim; ++d)
            if (d != f) rho.push_back(R[(size_t)r * dim + d]);
        rcomp[(size_t)r] = R[(size_t)r * dim + f];
        const bool negated = canonical_negate(rho);
        auto it = index.find(rho);
        if (it == index.end()) {
            it = index.emplace(rho, (int32_t)lists.size()).first;
            lists.emplace_back();
            plan.h_R2.insert(plan.h_R2.end(), rho.begin(), rho.end());
        }
        lists[(size_t)it->second].push_back((int32_t)r | (negated ? (int32_t)0x80000000 : 0));
    }
    plan.dim = dim;
    plan.n_r = n_r;
    plan.n_rho = (int64_t)lists.size();
    plan.k2 = (plan.n_rho * 2 + TBK_BK - 1) / TBK_BK * TBK_BK;
    plan.n_rho_pad = plan.k2 / 2;
    plan.capacity = capacity;
    std::vector<int64_t> lptr((size_t)plan.n_rho_pad + 1, 0);
    std::vector<int32_t> lrec;
    for (int64_t i = 0; i < plan.n_rho_pad; ++i) {
        if (i < plan.n_rho) lrec.insert(lrec.end(), lists[(size_t)i].begin(), lists[(size_t)i].end());
        lptr[(size_t)i + 1] = (int64_t)lrec.size();  // padding vectors keep empty lists: zero rows
    }
    std::vector<int32_t> r2((size_t)plan.n_rho_pad * std::max(dim - 1, 1), 0);
    std::copy(plan.h_R2.begin(), plan.h_R2.end(), r2.begin());
    const size_t row_len = (size_t)ncol_pad * 2;
    TBK_HIP(hipMalloc((void**)&plan.d_R2, std::max<size_t>(r2.size(), 1) * sizeof(int32_t)));
    TBK_HIP(hipMalloc((void**)&plan.d_lptr, lptr.size() * sizeof(int64_t)));
    TBK_HIP(hipMalloc((void**)&plan.d_lrec, std::max<size_t>(lrec.size(), 1) * sizeof(int32_t)));
    TBK_HIP(hipMalloc((void**)&plan.d_rcomp, std::max<size_t>(rcomp.size(), 1) * sizeof(int32_t)));
    TBK_HIP(hipMalloc((void**)&plan.d_B2, (size_t)capacity * plan.k2 * row_len * sizeof(double)));
    plan.table_entries = n_r * FOLD_GROUP;
    TBK_HIP(hipMalloc((void**)&plan.d_table, std::max<size_t>((size_t)plan.table_entries * 2, 1) * sizeof(double)));
    TBK_HIP(hipMemcpy(plan.d_R2, r2.data(), r2.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    TBK_HIP(hipMemcpy(plan.d_lptr, lptr.data(), lptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    TBK_HIP(hipMemcpy(plan.d_lrec, lrec.data(), lrec.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    TBK_HIP(hipMemcpy(plan.d_rcomp, rcomp.data(), rcomp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    plan.built = true;
    return TBK_OK;
}

int tbk_fold_plan(tbk_model* m, int f) {
    return build_plan(m->fold[f], m->h_R.data(), m->n_r, m->dim, f, m->ncol_pad, FOLD_GROUP);
}

// Second-level plan: folds the lattice `parent` produced along its component f2 (mesh lines inside a mesh plane);
// room for `capacity` operands (one per line of a plane piece).
int tbk_fold_subplan(tbk_model* m, tbk_fold_plan_t& parent, int f2, int capacity, tbk_fold_plan_t** out) {
    *out = nullptr;
    const int dim2 = parent.dim - 1;
    if (dim2 < 2 || f2 < 0 || f2 >= dim2) return TBK_OK;
    if (!parent.sub) parent.sub = new tbk_fold_plan_t[TBK_MAX_DIM];
    tbk_fold_plan_t& plan = parent.sub[f2];
    if (plan.built && plan.capacity < capacity) {  // a longer piece than any before: rebuild with more room
        void* ptrs[] = {plan.d_R2, plan.d_lptr, plan.d_lrec, plan.d_rcomp, plan.d_B2, plan.d_table};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        plan = tbk_fold_plan_t();
    }
    TBK_CHECK(build_plan(plan, parent.h_R2.data(), parent.n_rho, dim2, f2, m->ncol_pad, capacity));
    *out = &plan;
    return TBK_OK;
}

static void release_plan(tbk_fold_plan_t& plan) {
    if (plan.sub) {
        for (int i = 0; i < TBK_MAX_DIM; ++i) release_plan(plan.sub[i]);
        delete[] plan.sub;
    }
    void* ptrs[] = {plan.d_R2, plan.d_lptr, plan.d_lrec, plan.d_rcomp, plan.d_B2, plan.d_table};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    plan = tbk_fold_plan_t();
}

void tbk_fold_release(tbk_model* m) {
    for (tbk_fold_plan_t& plan : m->fold) release_plan(plan);
}

// Average run length from which folding pays: a run costs one pass over Bt (60 us at the headline shape) plus
// three small launches, the direct contraction ~1 us per k-point.  Measured on meshes of the headline model:
// 14^3 (runs of 196) 4.4 -> 2.4 ms, 20^3 11.4 -> 5.0 ms, 30^3 38 -> 13.6 ms, 50^3 159 -> 49 ms.
int64_t tbk_fold_min_run() { return 128; }

// Which component (if any) is worth folding for this k list: the one with the fewest runs of equal consecutive
// values, if its runs average >= tbk_fold_min_run() k-points and the folded lattice is at least 3x smaller.  -1: none.
int tbk_fold_choose(tbk_model* m, const double* h_k, int64_t nk, std::vector<int64_t>& run_starts) {
    run_starts.clear();
    if (!m->fold_enabled || m->sparse || m->kdotp || m->dim < 2 || m->n_r < 64 || nk < 1024 || m->h_R.empty()) return -1;
    int best = -1;
    int64_t best_runs = nk;
    for (int d = 0; d < m->dim; ++d) {
        int64_t runs = 1;
        for (int64_t i = 1; i < nk && runs * tbk_fold_min_run() <= nk; ++i)
            if (h_k[i * m->dim + d] != h_k[(i - 1) * m->dim + d]) ++runs;
        if (runs * tbk_fold_min_run() <= nk && runs < best_runs) {
            best_runs = runs;
            best = d;
        }
    }
    if (best < 0) return -1;
    if (tbk_fold_plan(m, best) != TBK_OK) return -1;
    if (m->fold[best].n_rho * 3 > m->n_r) return -1;
    run_starts.push_back(0);
    for (int64_t i = 1; i < nk; ++i)
        if (h_k[i * m->dim + best] != h_k[(i - 1) * m->dim + best]) run_starts.push_back(i);
    run_starts.push_back(nk);
    return best;
}

int tbk_fold_group_size() { return FOLD_GROUP; }

// Folds the CURRENT operand of `m` (m->d_B: the staged one, or a first-level folded one) for the n_g (<= FOLD_GROUP)
// shared-component values h_kf[] in one pass, into slots slot0 .. slot0 + n_g - 1 of the plan's buffer (main stream).
int tbk_fold_group(tbk_model* m, tbk_fold_plan_t& plan, const double* h_kf, int n_g, int slot0) {
    const int64_t row_len = (int64_t)m->ncol_pad * 2;
    double* out = plan.d_B2 + (size_t)slot0 * plan.k2 * row_len;
    StageTimer t(m, TBK_T_PHASE);
    if (n_g == 1) {
        dim3 grid((unsigned)((row_len + 255) / 256), (unsigned)plan.n_rho_pad);
        hipLaunchKernelGGL(fold_rows_kernel, grid, dim3(256), 0, m->stream, m->d_B, row_len, plan.d_lptr, plan.d_lrec,
                           plan.d_rcomp, h_kf[0], out);
        TBK_HIP(hipGetLastError());
        return TBK_OK;
    }
    FoldValues values;
    for (int g = 0; g < FOLD_GROUP; ++g) values.v[g] = g < n_g ? h_kf[g] : 0.0;
    const int64_t entries = plan.n_r * n_g;
    hipLaunchKernelGGL(fold_table_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, m->stream, plan.d_rcomp,
                       plan.n_r, values, n_g, plan.d_table);
    TBK_HIP(hipGetLastError());
    dim3 grid((unsigned)((row_len + 255) / 256), (unsigned)plan.n_rho_pad);
    hipLaunchKernelGGL(fold_rows_group_kernel, grid, dim3(256), 0, m->stream, m->d_B, row_len, plan.d_lptr, plan.d_lrec,
                       plan.d_table, n_g, plan.k2 * row_len, out);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

// All n_lines lines of a mesh plane piece in one go: their shared-component values are read on the device
// (d_kf[line * stride]); slots slot0 .. slot0 + n_lines - 1 of the plan's buffer (slot0 + n_lines <= plan.capacity).
int tbk_fold_lines(tbk_model* m, tbk_fold_plan_t& plan, const double* d_kf, int64_t stride, int n_lines, int slot0) {
    const int64_t row_len = (int64_t)m->ncol_pad * 2;
    StageTimer t(m, TBK_T_PHASE);
    if ((size_t)plan.table_entries < (size_t)plan.n_r * n_lines) {
        if (plan.d_table) TBK_HIP(hipFree(plan.d_table));
        plan.d_table = nullptr;
        plan.table_entries = plan.n_r * (int64_t)std::max(n_lines, plan.capacity);
        TBK_HIP(hipMalloc((void**)&plan.d_table, std::max<size_t>((size_t)plan.table_entries * 2, 1) * sizeof(double)));
    }
    const int64_t entries = plan.n_r * n_lines;
    hipLaunchKernelGGL(fold_table_dev_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, m->stream,
                       plan.d_rcomp, plan.n_r, d_kf, stride, n_lines, plan.d_table);
    TBK_HIP(hipGetLastError());
    dim3 grid((unsigned)((row_len + 255) / 256), (unsigned)plan.n_rho_pad, (unsigned)((n_lines + FOLD_GROUP - 1) / FOLD_GROUP));
    hipLaunchKernelGGL(fold_rows_group_kernel, grid, dim3(256), 0, m->stream, m->d_B, row_len, plan.d_lptr, plan.d_lrec,
                       plan.d_table, n_lines, plan.k2 * row_len, plan.d_B2 + (size_t)slot0 * plan.k2 * row_len);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

// Turns `m` into the model folded by `plan`, with the operand in `slot` of the plan's buffer; the caller evaluates and
// calls tbk_fold_leave.  Nests (a second-level plan on top of a first-level one).
int tbk_fold_enter(tbk_model* m, tbk_fold_plan_t& plan, int slot, tbk_fold_saved_t& saved) {
    const int64_t row_len = (int64_t)m->ncol_pad * 2;
    saved.dim = m->dim;
    saved.n_r = m->n_r;
    saved.n_r_pad = m->n_r_pad;
    saved.k2 = m->k2;
    saved.d_R = m->d_R;
    saved.d_B = m->d_B;
    m->dim = saved.dim - 1;
    m->n_r = plan.n_rho;
    m->n_r_pad = plan.n_rho_pad;
    m->k2 = plan.k2;
    m->d_R = plan.d_R2;
    m->d_B = plan.d_B2 + (size_t)slot * plan.k2 * row_len;
    return TBK_OK;
}

void tbk_fold_leave(tbk_model* m, const tbk_fold_saved_t& saved) {
    m->dim = saved.dim;
    m->n_r = saved.n_r;
    m->n_r_pad = saved.n_r_pad;
    m->k2 = saved.k2;
    m->d_R = saved.d_R;
    m->d_B = saved.d_B;
}

int tbk_fold_drop_component(tbk_model* m, const double* d_k, int dim, int f, int64_t nk, double* d_k2) {
    hipLaunchKernelGGL(drop_component_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, m->stream, d_k, dim, f, nk,
                       d_k2);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
