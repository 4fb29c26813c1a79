// HOST build of the BODY-PER-LANE simulator kernel (parc_amd/csrc/parc_sim_bpl.h, the kernel the product runs) -- TEST
// INFRASTRUCTURE ONLY.  The kernel is SPMD code: 16 lanes per env that exchange values through wave shuffles, a ballot and
// LDS between barriers.  Here every lane is a fiber (ucontext) and the 16 fibers of an env run in lock step: each lane
// primitive is a rendezvous.  Because all lanes execute the same sequence of primitives, a rendezvous is ONE switch to the next
// lane of the ring: lane 0 only gets control back from lane 15, i.e. after every lane has arrived.  The arithmetic is the
// header's own, so ASan / UBSan / the NaN-poison build and the CPU invariant tests see the product kernel's code, and its
// results can be compared with the one-env-per-lane core on the CPU (two independent formulations of the same equations).
#define PARC_LANE_EMU 1
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#endif

#define __device__
#define __forceinline__ inline

namespace lane_emu {
constexpr int LANES = 16;
constexpr size_t STACK_BYTES = 512 << 10;
struct Group {
    ucontext_t main_ctx, ctx[LANES];
    char *stack[LANES];
    int cur;                       // lane that is running
    float xf[LANES];               // exchange slots of the collectives
    int xi[LANES];
    void (*body)(int lane, void *arg);
    void *arg;
};
static thread_local Group *G = nullptr;

static inline void switch_to(ucontext_t *from, ucontext_t *to, const void *to_stack, size_t to_size) {
#if defined(__SANITIZE_ADDRESS__)
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, to_stack, to_size);
    swapcontext(from, to);
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
    (void)to_stack;
    (void)to_size;
    swapcontext(from, to);
#endif
}
// every lane has reached this point when the call returns
static inline void rendezvous() {
    Group *g = G;
    const int me = g->cur, nx = (me + 1) % LANES;
    g->cur = nx;
    switch_to(&g->ctx[me], &g->ctx[nx], g->stack[nx], STACK_BYTES);
}
static void entry() {
    Group *g = G;
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_finish_switch_fiber(nullptr, nullptr, nullptr);
#endif
    const int me = g->cur;
    g->body(me, g->arg);
    // lanes finish in ring order: lane 0 first (it is the first to pass the last rendezvous), lane 15 last
    if (me + 1 < LANES) {
        g->cur = me + 1;
        switch_to(&g->ctx[me], &g->ctx[me + 1], g->stack[me + 1], STACK_BYTES);
    } else {
        g->cur = -1;
        switch_to(&g->ctx[me], &g->main_ctx, nullptr, 0);
    }
}
// fiber stacks are allocated once per host thread and reused for every env (an allocation per env step costs more than the step)
struct StackPool {
    char *s[LANES] = {};
    ~StackPool() {
        for (int l = 0; l < LANES; ++l) free(s[l]);
    }
};
static thread_local StackPool pool;

static void run(void (*body)(int, void *), void *arg) {
    Group g;
    g.body = body;
    g.arg = arg;
    for (int l = 0; l < LANES; ++l) {
        if (!pool.s[l]) pool.s[l] = (char *)malloc(STACK_BYTES);
        g.stack[l] = pool.s[l];
        getcontext(&g.ctx[l]);
        g.ctx[l].uc_stack.ss_sp = g.stack[l];
        g.ctx[l].uc_stack.ss_size = STACK_BYTES;
        g.ctx[l].uc_link = nullptr;
        makecontext(&g.ctx[l], entry, 0);
    }
    G = &g;
    g.cur = 0;
    switch_to(&g.main_ctx, &g.ctx[0], g.stack[0], STACK_BYTES);
    G = nullptr;
}
}  // namespace lane_emu

// ---- the lane primitives the kernel source uses (width is always the 16-lane group)
struct EmuTid {
    int x = lane_emu::G->cur;      // lane within the env's group (one env per "workgroup" here)
};
#define threadIdx (EmuTid{})
static inline float __shfl(float v, int src, int /*width*/) {
    lane_emu::Group *g = lane_emu::G;
    g->xf[g->cur] = v;
    lane_emu::rendezvous();
    const float r = lane_emu::G->xf[src & (lane_emu::LANES - 1)];
    lane_emu::rendezvous();                    // nobody overwrites a slot before everybody has read
    return r;
}
static inline int __shfl_xor(int v, int mask, int /*width*/) {
    lane_emu::Group *g = lane_emu::G;
    const int me = g->cur;
    g->xi[me] = v;
    lane_emu::rendezvous();
    const int r = lane_emu::G->xi[(me ^ mask) & (lane_emu::LANES - 1)];
    lane_emu::rendezvous();
    return r;
}
static inline unsigned long long __ballot(int pred) {
    lane_emu::Group *g = lane_emu::G;
    g->xi[g->cur] = pred ? 1 : 0;
    lane_emu::rendezvous();
    unsigned long long m = 0ull;
    for (int l = 0; l < lane_emu::LANES; ++l) m |= (unsigned long long)lane_emu::G->xi[l] << l;
    lane_emu::rendezvous();
    return m;
}
static inline void __syncthreads() { lane_emu::rendezvous(); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }

#include "../parc_amd/csrc/parc_sim_bpl.h"

namespace {
struct EnvArgs {
    const parc_sim_model_t *m;
    const parc_terrain_t *ter;
    float *root_state, *dof_state, *rigid_body_state, *contact_forces;
    const float *env_offset, *action, *lo, *hi;
    int n_sub;
    float h;
    float *lds, *cc;
};
void lane_body(int lane, void *p) {
    EnvArgs &a = *(EnvArgs *)p;
    parc_sim_bpl::step_lane(*a.m, *a.ter, lane, a.root_state, a.dof_state, a.rigid_body_state, a.contact_forces, a.env_offset, a.action,
                            a.lo, a.hi, a.n_sub, a.h, a.lds, a.cc + lane * (BPL_CC_SLOTS * BPL_CC_FLOATS + 1));
}
int g_fill = -1;        // >= 0: LDS and the contact cache start from this byte pattern (0xFF = NaN) instead of zero
}  // namespace

extern "C" void sim_host_bpl_set_fill(int byte) { g_fill = byte; }

extern "C" int sim_host_step_bpl(const parc_sim_model_t *model, parc_terrain_t terrain, int n_envs, float *root_state, float *dof_state,
                                 float *rigid_body_state, float *contact_forces, const float *env_offsets, const float *action,
                                 const float *action_low, const float *action_high, int n_substeps, float h) {
    const int B = model->num_bodies, D = model->dof_size;
    if (B > lane_emu::LANES) return -2;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n_envs; ++e) {
        float lds[BPL_G * BPL_CONTRIB];
        float cc[lane_emu::LANES * (BPL_CC_SLOTS * BPL_CC_FLOATS + 1)];
        memset(lds, g_fill < 0 ? 0 : g_fill, sizeof lds);          // (device LDS is not initialised either)
        memset(cc, g_fill < 0 ? 0 : g_fill, sizeof cc);
        EnvArgs a{model, &terrain, root_state + 13 * (size_t)e, dof_state + 2 * (size_t)D * e, rigid_body_state + 13 * (size_t)B * e,
                  contact_forces + 3 * (size_t)B * e, env_offsets + 3 * (size_t)e, action + (size_t)D * e, action_low, action_high,
                  n_substeps, h, lds, cc};
        lane_emu::run(lane_body, &a);
    }
    return 0;
}
