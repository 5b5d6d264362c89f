// tbk_multi.hip -- one call, several devices: the k list of a host-buffer call is cut into contiguous slabs, one per
// staged handle, and every slab runs on its handle's device from its own host thread.
//
// The reference evaluates k-points one after the other in ONE process (/root/reference/src/tbmodels/_tb_model.py:1111-1123
// has no cross-k term, :1147-1150 returns rows in caller order); this is the form of the sharded path that keeps that
// surface: Model.eigenval / Model.hamilton stay single calls of a single process, no launcher, no process group.
// With host output there is nothing to exchange -- every device copies its slab straight into its rows of the caller's
// array -- so no collective is issued here (the RCCL all-gather of tbk_comm.hip serves device-resident consumers).
//
// Slabs are the same contiguous ceil(nk / n) rows that tbmodels_amd.sharding.slab_bounds gives the ranks of the
// one-process-per-GPU form, so both forms evaluate a k-point on the same device index and fold mesh slabs alike.

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "tbk_internal.h"

namespace {

struct SlabResult {
    int status = TBK_OK;
    std::string message;
};

template <class Call>
int run_slabs(int n, int64_t nk, Call&& call) {  // call(i, lo, count): slab i on handle i
    const int64_t per = (nk + n - 1) / n;
    // slabs that hold k-points: a one-k call under TBK_DEVICES=0..7 has ONE, and must not pay seven thread starts
    const int busy = (int)std::min<int64_t>(n, (nk + per - 1) / std::max<int64_t>(per, 1));
    if (busy <= 1) return call(0, 0, nk);
    std::vector<SlabResult> results((size_t)busy);
    auto work = [&](int i) {
        const int64_t lo = std::min<int64_t>(nk, (int64_t)i * per), hi = std::min<int64_t>(nk, lo + per);
        if (hi <= lo) return;
        const int status = call(i, lo, hi - lo);
        results[(size_t)i].status = status;
        if (status != TBK_OK) results[(size_t)i].message = tbk_last_error();  // this thread's message
    };
    std::vector<std::thread> threads;
    threads.reserve((size_t)busy);
    try {
        for (int i = 1; i < busy; ++i) threads.emplace_back(work, i);
    } catch (...) {
        for (auto& t : threads) t.join();
        tbk_set_error("cannot start a host thread per device");
        return TBK_ERR_DEVICE;
    }
    work(0);  // the caller's thread takes the first slab
    for (auto& t : threads) t.join();
    // the failure of the FIRST failing slab in k order is the call's failure: what a loop over the k list would have hit first
    for (int i = 0; i < busy; ++i) {
        if (results[(size_t)i].status != TBK_OK) {
            tbk_set_error("%s", results[(size_t)i].message.c_str());
            return results[(size_t)i].status;
        }
    }
    return TBK_OK;
}

int check_handles(tbk_model* const* handles, int n) {
    TBK_ARG(handles != nullptr && n >= 1, "no handles");
    for (int i = 0; i < n; ++i) {
        TBK_ARG(handles[i] != nullptr, "a handle is NULL");
        TBK_ARG(handles[i]->dim == handles[0]->dim && handles[i]->n_orb == handles[0]->n_orb,
                "handles of different models (dim / n_orb differ)");
    }
    return TBK_OK;
}

}  // namespace

extern "C" int tbk_eigenval_multi(tbk_model* const* handles, int n_handles, const double* k, int64_t nk, double* E_out) {
    TBK_CHECK(check_handles(handles, n_handles));
    TBK_ARG(nk >= 0, "nk < 0");
    if (nk == 0) return TBK_OK;
    TBK_ARG(k && E_out, "k / E is NULL");
    const int dim = handles[0]->dim, n_orb = handles[0]->n_orb;
    if (n_handles == 1) return tbk_eigenval(handles[0], k, nk, E_out);
    return run_slabs(n_handles, nk, [&](int i, int64_t lo, int64_t count) {
        return tbk_eigenval(handles[i], k + lo * dim, count, E_out + lo * n_orb);
    });
}

extern "C" int tbk_hamilton_multi(tbk_model* const* handles, int n_handles, const double* k, int64_t nk, int convention,
                                  const double* pos, double* H_out) {
    TBK_CHECK(check_handles(handles, n_handles));
    TBK_ARG(convention == 1 || convention == 2, "convention must be 1 or 2");
    TBK_ARG(nk >= 0, "nk < 0");
    if (nk == 0) return TBK_OK;
    TBK_ARG(k && H_out, "k / H is NULL");
    TBK_ARG(convention == 2 || pos != nullptr, "convention 1 needs pos");
    const int dim = handles[0]->dim;
    const int64_t nn2 = (int64_t)handles[0]->n_orb * handles[0]->n_orb * 2;
    if (n_handles == 1) return tbk_hamilton(handles[0], k, nk, convention, pos, H_out);
    return run_slabs(n_handles, nk, [&](int i, int64_t lo, int64_t count) {
        return tbk_hamilton(handles[i], k + lo * dim, count, convention, pos, H_out + lo * nn2);
    });
}

// k.p models on several devices (kdotp.py:51-100 has the same two methods as Model): the same slabs, through the k.p
// entry points of every staged copy
extern "C" int tbk_kdotp_eigenval_multi(tbk_kdotp* const* handles, int n_handles, const double* k, int64_t nk, double* E_out) {
    TBK_ARG(handles != nullptr && n_handles >= 1, "no handles");
    std::vector<tbk_model*> cores((size_t)n_handles);
    for (int i = 0; i < n_handles; ++i) {
        TBK_ARG(handles[i] != nullptr && handles[i]->core != nullptr, "a handle is NULL");
        cores[(size_t)i] = handles[i]->core;
    }
    return tbk_eigenval_multi(cores.data(), n_handles, k, nk, E_out);
}

extern "C" int tbk_kdotp_hamilton_multi(tbk_kdotp* const* handles, int n_handles, const double* k, int64_t nk, double* H_out) {
    TBK_ARG(handles != nullptr && n_handles >= 1, "no handles");
    std::vector<tbk_model*> cores((size_t)n_handles);
    for (int i = 0; i < n_handles; ++i) {
        TBK_ARG(handles[i] != nullptr && handles[i]->core != nullptr, "a handle is NULL");
        cores[(size_t)i] = handles[i]->core;
    }
    return tbk_hamilton_multi(cores.data(), n_handles, k, nk, 2, nullptr, H_out);
}
