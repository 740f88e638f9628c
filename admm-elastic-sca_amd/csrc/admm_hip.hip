// admm_hip.hip -- the device translation unit of libadmm_hip.so: kernels (kernels_*.hpp, factor_dev.hpp), numeric factorization on the
// device, upload, launches and the step loop, and the C ABI around them.  Host-only parts: host_setup.cpp, partition.cpp, comm.cpp (ctx.hpp).
// See include/admm_hip.h for the contract and DESIGN.md for the design.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "force_init.hpp"
#include "kernels_global.hpp"
#include "factor_dev.hpp"
#include "kernels_local.hpp"

extern "C" int omp_get_max_threads(void);

using namespace admm_host;
using admm_dev::BatchDev;
using admm_dev::FactorDev;

using namespace admm_lib;

namespace {

// captured iterations carry device addresses and flags in their kernel arguments: whatever changes those drops the graphs
void drop_iteration_graphs(admm_hip_ctx *ctx) {
    if (ctx->iter_exec) { (void)hipGraphExecDestroy(ctx->iter_exec); ctx->iter_exec = nullptr; }
    if (ctx->iter_graph) { (void)hipGraphDestroy(ctx->iter_graph); ctx->iter_graph = nullptr; }
    if (ctx->frame_exec) { (void)hipGraphExecDestroy(ctx->frame_exec); ctx->frame_exec = nullptr; }
    if (ctx->frame_graph) { (void)hipGraphDestroy(ctx->frame_graph); ctx->frame_graph = nullptr; }
    ctx->frame_iters = 0;
}

void free_device(admm_hip_ctx *ctx) {
    drop_iteration_graphs(ctx);
    for (void *p : ctx->allocs) (void)hipFree(p);
    ctx->allocs.clear();
    for (Batch &b : ctx->batches) {
        if (b.h_tg) (void)hipHostFree(b.h_tg);
        if (b.h_ac) (void)hipHostFree(b.h_ac);
        if (b.upd_ev) (void)hipEventDestroy(b.upd_ev);
        b.h_tg = nullptr; b.h_ac = nullptr; b.upd_ev = nullptr;
    }
    ctx->levels.clear(); ctx->levels_top.clear();
}

#include "dev_factorize.inc"
#include "upload.inc"
#include "launch.inc"

} // namespace

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

#include "abi_setup.inc"
#include "abi_step.inc"
#include "abi_parity.inc"

} // extern "C"
