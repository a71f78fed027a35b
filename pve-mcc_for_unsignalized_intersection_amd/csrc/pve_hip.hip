// pve_hip.hip -- gfx950 (MI355X / CDNA4) kernels + HIP backend of the C ABI -> libpveenv.so
//
//   k_tick<CAP>     one workgroup (CAP = 64 or 128 threads = 1 or 2 wave64) per intersection:
//                   fused step* + scene_update + delete_vehicle, state staged in LDS, phases of
//                   pve_tick_core.h separated by workgroup barriers.
//   k_rollout<CAP>  pve_step_many: the same phases in a loop, the state resident on the chip between ticks
//   k_tick_geo<CAP> the 4- / 8-lane layouts (pve_tick_geo.h)
//   k_compact<CAP>  delete_vehicle() alone (split protocol of the single-env compat class)
//   k_reset<CAP>    constructor warm-up, one thread per intersection
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no FMA contraction: the discrete
// decisions of the reference sit on margins down to 4e-16).
#include <hip/hip_runtime.h>
#include <mutex>
#include <new>

#include "pve_host.h"
#include "pve_tick_core.h"
#include "pve_tick_geo.h"
#include "pve_actor.h"

using namespace pve;

// A/B knobs of the measurement tooling (tools/ab_launch_shapes.py, tools/gpu.sh): environment variables that switch launch
// shapes.  Compiled in only with -DPVE_AB_KNOBS (`make knobs` -> build/libpveenv_knobs.so); the product library reads no
// environment variable on its launch path.
#ifdef PVE_AB_KNOBS
#define PVE_KNOB(name_) getenv(name_)
#else
#define PVE_KNOB(name_) ((const char *)nullptr)
#endif
// the HOME build of the persistent table-source kernel (k_rollout<128, 5, ..>, 10 workgroups per CU) is what launch_rollout
// takes when it is eligible; -DPVE_HOME_DEFAULT=0 builds a library that keeps the 8-workgroup kernel (A/B)
#ifndef PVE_HOME_DEFAULT
#define PVE_HOME_DEFAULT 1
#endif

typedef __attribute__((address_space(4))) const char *KernargPtr;
// diagnostics (pve_debug_phase_cycles): every wave keeps the clock ticks it spent in each phase (incl. the
// wait at the closing barrier) in registers and adds them to its private row of a [n_waves][16] device
// buffer at the end of the kernel (no shared atomics, no extra memory traffic inside the phases)
#define PVE_PHASE_MARK(idx)                                                              \
    if (P.phase_cycles) {                                                                \
        unsigned long long now_ = wall_clock64();                                        \
        pc_[idx] = now_ - tprev_;                                                        \
        tprev_ = now_;                                                                   \
    }

// The workgroup barrier between two phases.  __syncthreads() = workgroup-scope release + s_barrier + acquire; on gfx950
// outside threadgroup-split mode this compiles to `s_waitcnt lgkmcnt(0)` + `s_barrier` (checked in the ISA): LDS accesses
// are ordered, and so are the global accesses of the workgroup's waves (one CU, one in-order vector-memory path) WITHOUT
// a vmcnt(0) -- prefetches and the output stores of one tick stay in flight across the phases of the next.
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
// CAP = 128: 5 waves per SIMD (<= 96 VGPR) + the 15.1 KB LDS block = 10 workgroups of 128 threads per CU instead of 8: with
// the sub-batches pipelined on several streams the tick scales almost linearly with the resident workgroups.
// CAP = 64: one wave per workgroup, LDS (10 KB) admits 16 workgroups per CU = 4 waves per SIMD (<= 128 VGPR).
template <int CAP>
__global__ __launch_bounds__(CAP) __attribute__((amdgpu_waves_per_eu(CAP == 64 ? 4 : 5, CAP == 64 ? 4 : 5))) void k_tick(const Const c_arg, const Params P_arg)
{
    KernargPtr ka0_ = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const PVE_AS4 Const &c = *(const PVE_AS4 Const *)ka0_;
    const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(ka0_ + ((sizeof(Const) + 7) / 8 * 8));
    typedef Shared<CAP, false> ShT;                  // (without the lane << 16 | j words: 16 368 B, the 10th workgroup per CU)
    __shared__ ShT sh;
    const int t = threadIdx.x;
    const int env = blockIdx.x;
    Regs r;
    typedef Tick<CAP, ShT> T;
    unsigned long long pc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev_ = P.phase_cycles ? wall_clock64() : 0ull;
    T::ph_load(c, P, env, t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(0)
    if (P.stop_phase == 0) return;          // (diagnostics, uniform: pve_debug_stop_phase)
    T::ph_step1(c, P, env, t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(1)
    if (P.stop_phase == 1) return;          // (diagnostics, uniform: pve_debug_stop_phase)
    T::ph_step2(c, t, sh, r);
    T::ph_lists_a(c, t, sh);
    lds_barrier();
    PVE_PHASE_MARK(2)
    if (P.stop_phase == 2) return;          // (diagnostics, uniform: pve_debug_stop_phase)
    T::ph_step3(c, t, sh, r);
    T::ph_step3_publish(t, sh, r);
    T::ph_lists_b(t, sh);
    lds_barrier();
    PVE_PHASE_MARK(3)
    if (P.stop_phase == 3) return;          // (diagnostics, uniform: pve_debug_stop_phase)
    T::ph_build(c, t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(4)
    if (P.stop_phase == 4) return;          // (diagnostics, uniform: pve_debug_stop_phase)
    T::ph_rank(t, sh, env);
    lds_barrier();
    PVE_PHASE_MARK(5)
    if (P.stop_phase == 5) return;          // (diagnostics, uniform: pve_debug_stop_phase)
    T::ph_scan(c, t, sh, r);
    PVE_PHASE_MARK(11)
    T::ph_reward(c, t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(6)
    if (P.stop_phase == 6) return;          // (diagnostics, uniform: pve_debug_stop_phase)
    T::ph_effects(c, t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(7)
    if (P.stop_phase == 7) return;          // (diagnostics, uniform: pve_debug_stop_phase)
    T::ph_prefetch_arrival(P, env, t, sh, r, NL);
    T::ph_lock(c, t, sh, r);
    lds_barrier();
    T::ph_lock2(t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(8)
    if (P.stop_phase == 8) return;          // (diagnostics, uniform: pve_debug_stop_phase)
    T::ph_final(c, P, env, t, sh, r);
    PVE_PHASE_MARK(9)
    if (P.out.state_pre) {            // uniform branch: optional training output
        __threadfence_block();
        __syncthreads();
        T::ph_state_publish(P.out, t, sh, r);
        __syncthreads();
        T::ph_state_coop(P, P.out, env, t, sh);
        PVE_PHASE_MARK(10)
    }
    if (P.phase_cycles && (t & 63) == 0) {
        unsigned long long *row = P.phase_cycles + ((size_t)env * (CAP / 64) + (t >> 6)) * 16;
#pragma unroll
        for (int k = 0; k < 12; k++) row[k] += pc_[k];
    }
}

// k_rollout<.., ACT>: the actor pass of one resident tick (behind STAGE: the carried state lives in the staging arrays, the
// registers are free).  The controlled vehicles are the dense threads 0 .. n_ctl-1 -- normally all of them in wave 0, whose
// chain is the workgroup's critical path -- so the work is split by TILE, not by owner: tile i = dense threads 32 i .. 32 i
// + 31 goes to wave i % (CAP / 64).  A tile's rows are the float32 images of what FIN stored to obs_post a moment ago (same
// CU: L1 / L2 hot, ordered by barrier A), addressed through the post-compaction slots the dense threads left in `adsts`
// (255 = the vehicle is gone); lane (j, hf) fetches the half row it contracts.  act[slot] receives the action of every
// controlled vehicle that keeps a slot.  The launch's first tick uses the same routine on the rows in HBM.
template <int CAP>
__device__ __forceinline__ void rollout_actor(const unsigned char *packed, const float *aprm, double *act, const uint8_t *adsts,
                                              int n_ctl, const void *rows, bool obs_f32, size_t base, int t)
{
    const pve_v8h *A1 = (const pve_v8h *)(packed + AP_A1), *A2 = (const pve_v8h *)(packed + AP_A2);
    const int lane = t & 63, hf = lane >> 5, j = lane & 31;
    for (int tile = t >> 6; 32 * tile < n_ctl; tile += CAP / 64) {          // (uniform per wave)
        const int slot = adsts[32 * tile + j];
        const int srow = slot == 255 ? 0 : slot;
        float x[16];
        if (obs_f32) actor_fetch_env((const float *)rows + base * OBSW, srow, hf, x);
        else actor_fetch_env((const double *)rows + base * OBSW, srow, hf, x);
        const float a = actor_tile32(A1, A2, aprm, x, lane);
        if (lane < 32 && slot != 255) act[slot] = (double)a;
    }
}

// PVE_SRC_TABLE: table[row][min(id, table_ids - 1)] -- a UNIFORM 64-bit row base plus a 32-bit lane offset (the `saddr + voffset`
// form of a global load; the plain index is a sign extension + 64-bit multiply-add per lane); the unsigned minimum also clamps a
// negative id (an empty slot, whose value is never used)
__device__ __forceinline__ double table_action(const double *table, int row, int table_ids, int id)
{
    const unsigned col = (unsigned)id < (unsigned)table_ids ? (unsigned)id : (unsigned)(table_ids - 1);
    return *(const double *)((const char *)(table + (size_t)row * (size_t)table_ids) + col * 8u);
}

// ------------------------------------------------------------------ persistent roll-out: the work queue
// One launch for a whole pve_step_many call: as many workgroups as the chip holds at once, each pulling (intersection,
// chunk) items -- chunk c of intersection e = ticks [c T, (c + 1) T) of the call -- until the call is done.  Nothing waits in
// launch order for the slowest intersection of a chunk; an item only waits for the previous chunk of ITS intersection.
//
// Hand-off of an intersection between two workgroups (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement &
// inter-workgroup visibility"): the eight XCDs' L2s are not coherent with each other and a CU's vector L1 is never
// refreshed by another CU's stores.  So (1) every intersection is worked by ONE XCD for the whole launch: env e belongs to
// shard e % n_shards, and a shard belongs to the first XCD -- HW_REG_XCC_ID as read by the executing wave, not an
// assumption about dispatch -- whose workgroup claims it (compare-and-swap on owner[]; an XCD starts with the shard of
// its own number and adopts what nobody has claimed once that is exhausted, so every shard is worked whatever the
// placement).  All stores to an intersection's state AND to its output rows (which successive ticks overwrite) then
// meet in one L2, in program order of the workgroups that are ordered by the done[] hand-off; (2) the finishing workgroup
// drains its stores (`s_waitcnt vmcnt(0)` in every wave: the write-through L1 has handed them to the L2), joins at a
// barrier and publishes done[e] with an agent-scope atomic store; (3) the next workgroup polls done[e] with agent-scope
// atomic loads (one lane, `s_sleep` between polls), joins at a barrier and reads the state with agent-scope atomic
// loads (`sc1`: served by the L2, never by its own L1).  Queue heads and done[] are agent-scope atomics throughout.
// Items of a shard are handed out chunk-major (all intersections' chunk c before any chunk c + 1) by a returning
// atomic add, so whoever holds an item is a RUNNING workgroup and the item it may wait for was handed out earlier:
// the oldest unfinished item never waits, whatever the grid size or whoever shares the chip.
// diagnostics build only (make EXTRA=-DPVE_QUEUE_TRACE, tools/persistent_trace.py): per-item timestamps; compiled out of the
// product library (the extra live values cost the one-wave variant 32 spilled registers)
#ifdef PVE_QUEUE_TRACE
#define Q_TRACE(R_) ((R_).q_trace)
#else
#define Q_TRACE(R_) ((unsigned long long *)nullptr)
#endif
__device__ __forceinline__ unsigned q_xcc_id()
{
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xFu;      // HW_REG_XCC_ID[3:0] (gfx942 / gfx950)
}
__device__ __forceinline__ unsigned q_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void q_store(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane: the next item of this workgroup's XCD -> (env, chunk), env < 0 when the call has no work left for it.
// `shard` / `probe` persist across calls (the shard being worked, the number of shards this workgroup has looked at).
__device__ __forceinline__ void q_dequeue(RolloutQueue *Q, int n_shards, int n_envs, int n_chunks, unsigned xcc, int &shard, int &probe,
                                          int &env, int &chunk)
{
    for (;;) {
        if (shard >= 0) {
            const unsigned es = (unsigned)(n_envs - shard + n_shards - 1) / (unsigned)n_shards;   // intersections of the shard
            const unsigned i = __hip_atomic_fetch_add(&Q->s[shard].head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i < es * (unsigned)n_chunks) {
                const unsigned c = i / es;
                env = shard + (int)(i - c * es) * n_shards; chunk = (int)c;
                return;
            }
        }
        // the shard is exhausted (or this is the first call): the next shard this XCD may work -- its own number first.
        // Loads before atomics: at the end of a call every workgroup comes through here, and a compare-and-swap per
        // workgroup and shard on lines the running workgroups' XCDs share is a flood (measured: the last items of a call ran
        // 1.4x slower under it)
        shard = -1;
        while (probe < n_shards) {
            const int cand = (int)((xcc + (unsigned)probe) % (unsigned)n_shards);
            probe++;
            if (cand >= n_envs) continue;
            unsigned seen = q_load(&Q->s[cand].owner);
            if (seen == 0u)
                __hip_atomic_compare_exchange_strong(&Q->s[cand].owner, &seen, xcc + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen != 0u && seen != xcc + 1u) continue;          // another XCD's
            const unsigned es = (unsigned)(n_envs - cand + n_shards - 1) / (unsigned)n_shards;
            if (q_load(&Q->s[cand].head) >= es * (unsigned)n_chunks) continue;      // ours, but handed out already
            shard = cand;
            break;
        }
        if (shard < 0) { env = -1; chunk = 0; return; }
    }
}

// The workgroup's next item: lane 0 pulls it and waits for the previous chunk of its intersection (another workgroup's item);
// everybody learns it through q_word (LDS).  -> false when the call has no work left for this workgroup's XCD.
// q_word[0..1] = (env, chunk), [2..3] = lane 0's dequeue state (shard, shards looked at).
__device__ __forceinline__ bool q_take(const PVE_AS4 RolloutArgs &R, int n_envs, int t0, int *q_word, int &env, int &chunk, int &k_base,
                                       int &n_ticks)
{
    RolloutQueue *Q = (RolloutQueue *)R.queue;
    unsigned *done = R.queue + sizeof(RolloutQueue) / 4;
    if (t0 == 0) {
        int shard = q_word[2], probe = q_word[3], e, ch;
        const unsigned long long tq0 = Q_TRACE(R) ? wall_clock64() : 0ull;
        q_dequeue(Q, R.n_shards, n_envs, R.n_full + R.n_taper, q_xcc_id(), shard, probe, e, ch);
        q_word[0] = e; q_word[1] = ch; q_word[2] = shard; q_word[3] = probe;
        const unsigned long long tq1 = Q_TRACE(R) ? wall_clock64() : 0ull;
        if (e >= 0)                                   // the previous chunk of this intersection (another workgroup's item)
            while ((int)(q_load(&done[e]) - (R.done_base + (unsigned)ch)) < 0) __builtin_amdgcn_s_sleep(8);
        if (Q_TRACE(R) && e >= 0) {                    // diagnostics: dequeue start, item known, predecessor done, who
            unsigned long long *row = Q_TRACE(R) + ((size_t)ch * n_envs + e) * 8;
            row[0] = tq0; row[1] = tq1; row[2] = wall_clock64();
            row[6] = (unsigned long long)blockIdx.x | ((unsigned long long)q_xcc_id() << 32);
        }
    }
    __syncthreads();
    env = __builtin_amdgcn_readfirstlane(q_word[0]);
    chunk = __builtin_amdgcn_readfirstlane(q_word[1]);
    if (env < 0) return false;
    rollout_item(R, chunk, k_base, n_ticks);
    return true;
}
// hand the intersection on: every wave's stores (state, header, the ticks' output rows) have reached the L2, then the count of
// completed items of this intersection goes up by one
__device__ __forceinline__ void q_publish(const PVE_AS4 RolloutArgs &R, int n_envs, int t0, int env, int chunk)
{
    const unsigned long long tf0 = (Q_TRACE(R) && t0 == 0) ? wall_clock64() : 0ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t0 == 0) q_store(R.queue + sizeof(RolloutQueue) / 4 + env, R.done_base + (unsigned)chunk + 1u);
    if (Q_TRACE(R) && t0 == 0) {                      // diagnostics: state flushed (stores issued), handed on
        unsigned long long *row = Q_TRACE(R) + ((size_t)chunk * n_envs + env) * 8;
        row[4] = tf0; row[5] = wall_clock64();
    }
}
// the last workgroup to leave clears the queue words for the next launch (done[] stays: it is cumulative)
__device__ __forceinline__ void q_leave(const PVE_AS4 RolloutArgs &R, int t0)
{
    RolloutQueue *Q = (RolloutQueue *)R.queue;
    if (t0 == 0) {
        const unsigned left = __hip_atomic_fetch_add(&Q->exits, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left + 1u == gridDim.x) {
            for (int k = 0; k < QUEUE_MAX_SHARDS; k++) { q_store(&Q->s[k].head, 0u); q_store(&Q->s[k].owner, 0u); }
            q_store(&Q->exits, 0u);
        }
    }
}

// pve_step_many: R.n_ticks ticks of one intersection per workgroup, the state resident in registers / LDS between the
// ticks (pve_tick_core.h, "k_rollout").  Per tick only the outputs go to HBM; the action of the next tick (pool) and
// the next arrival times are prefetched under the tail of the current one.
// WPE = waves per SIMD the register allocation is held to: 4 (<= 128 VGPR, no spills: 8 workgroups of 128 threads per CU,
// i.e. 4096 intersections in exactly two rounds) or 5 (<= 96 VGPR, 10 per CU for stream-pipelined sub-batches).
// PROF: the diagnostics build (pve_debug_phase_cycles) accumulates per-phase clock ticks over the ticks of the launch.
// ACT: pve_step_many(PVE_SRC_ACTOR) -- the closed loop of main.py:398-441 resident on the chip.  The dense thread of every
// controlled vehicle keeps the float32 observation row it has just built in FIN; behind STAGE (the carried state lives in
// the staging arrays by then, the registers are free) the wave runs the actor on its <= 2 tiles of 32 vehicles
// (pve_actor.h: actor_tile32, the same function the stand-alone k_actor_h calls), A operands streamed from L1 / L2, the
// float parameters in 2 KB of LDS, and deposits next tick's actions where RELOAD looks for them (act_next).  No new
// barrier; still ticks are staged like the others.  The first tick of a launch takes its rows from HBM.
// TRAIN: the training outputs (obs_pre, state_pre: SURVEY 8 f3) are written per tick; a variant of its own so that the
// default kernel keeps its register allocation (123 VGPR, no scratch).
// IDT: PVE_SRC_TABLE -- the action is a function of (tick, vehicle id) given as a table: every vehicle's own thread gathers
// its next action under FX (the id travels with the vehicle), the lanes that spawn gather theirs right behind FIN (in flight
// under barrier A and STAGE), and the values are parked at the vehicles' NEW slots (act_next) where RELOAD looks for them.
// PERS: the persistent form -- the workgroup pulls (intersection, chunk) items from the queue above; n_ticks = ticks per item.
template <int CAP, int WPE, bool PROF = false, bool ACT = false, bool TRAIN = false, bool IDT = false, bool PERS = false>
__global__ __launch_bounds__(CAP) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_rollout(const Const c_arg, const Params P_arg,
                                                                                               const RolloutArgs R_arg)
{
    static_assert(!PERS || !PROF, "the persistent form has no phase-cycle diagnostics variant");
    static_assert(!(ACT && IDT), "one action source per variant");
    KernargPtr ka0_ = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t OFF_P = (sizeof(Const) + 7) / 8 * 8, OFF_R = OFF_P + (sizeof(Params) + 7) / 8 * 8;
    // (WPE = 5, the 96-VGPR experiment: the block without the lane << 16 | j words is 16 368 B -> 10 workgroups per CU)
    // (WPE = 5 also: HOME -- the carried per-slot fields live in LDS instead of registers, pve_tick_core.h)
    // (the HOME block without the lane << 16 | j words: with them it is 16 896 B = 9 workgroups per CU, measured 24.75 against
    //  24.78 us -- 9 and 10 resident workgroups per CU run at the same rate, the words buy nothing there)
    typedef Shared<CAP, (CAP == 128 && WPE == 4), (WPE == 5)> ShT;
    __shared__ ShT sh;
    __shared__ __attribute__((aligned(64))) float aprm[ACT ? PV_TOTAL : 1];
    __shared__ uint8_t adsts[ACT ? CAP : 1];         // post-compaction slot of every dense thread's vehicle (255: gone)
    __shared__ int q_word[4];                       // PERS: item (env, chunk) of the workgroup + the dequeue state of lane 0
    int adst = -1;
    int t0_ = threadIdx.x;
    int env0_ = PERS ? 0 : blockIdx.x;
    int k_base_ = 0, chunk_ = 0;                     // PERS: first tick of the item within the call, its chunk number
    bool aprm_staged = false;                        // PERS + ACT: the float parameters are in LDS (once per workgroup)
    // the first wave carries the dense-mapped phases (the critical chain of the workgroup), the second one mostly waits at
    // the barriers: the first wave gets the issue slots first (30.0 -> 29.6 us per tick; not in k_tick, where the closed
    // loop's actor kernel shares the chip and loses more than the tick gains)
    // (round 6, the HOME build with its 5 waves per SIMD: the second wave at 2 instead of 0 -- whatever it runs is what the first wave
    //  waits for at the next barrier: 22.83 vs 22.95 us steady; 3 / 3 and 2 / 3 are worse: 23.1 / 23.45)
    if (CAP > 64) { if (__builtin_amdgcn_readfirstlane(t0_) < 64) __builtin_amdgcn_s_setprio(3); else if (WPE == 5) __builtin_amdgcn_s_setprio(2); }
    KernargPtr kav_ = ka0_;
    Regs r;
    FinCarry fc;
    typedef Tick<CAP, ShT> T;
    unsigned long long pc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev_ = PROF ? wall_clock64() : 0ull;
    const unsigned long long sclk0_ = PROF ? clock64() : 0ull, wclk0_ = tprev_;   // shader clock vs the 100 MHz constant clock
#undef PVE_PHASE_MARK
#define PVE_PHASE_MARK(idx)                                                              \
    if (PROF) {                                                                          \
        unsigned long long now_ = wall_clock64();                                        \
        pc_[idx] += now_ - tprev_;                                                       \
        tprev_ = now_;                                                                   \
    }
    int pool_idx, n_ticks;
    if constexpr (PERS) { if (t0_ == 0) { q_word[2] = -1; q_word[3] = 0; } }
  for (;;) {                                        // PERS: one pass per item; else exactly one pass
    {
        const PVE_AS4 Const &c = *(const PVE_AS4 Const *)ka0_;
        const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(ka0_ + OFF_P);
        const PVE_AS4 RolloutArgs &R = *(const PVE_AS4 RolloutArgs *)(ka0_ + OFF_R);
        pool_idx = R.pool_tick0;
        n_ticks = R.n_ticks;
        const double *act0 = nullptr;
        if constexpr (PERS) {
            int e, ch, kb, nt;
            if (!q_take(R, P.n_envs, t0_, q_word, e, ch, kb, nt)) break;
            env0_ = e; chunk_ = ch; k_base_ = kb; n_ticks = nt;
            if (R.source == 1 || R.source == 3) pool_idx = (R.pool_tick0 + k_base_) % R.n_pool;
            if (R.source == 1) act0 = R.pool + (size_t)pool_idx * (size_t)P.n_envs * CAP;
        }
        // PERS + ACT: only the FIRST item of an intersection in a call computes its first actions from the caller's rows; every
        // later item takes them from `actor_actions`, where the previous item left them (its last tick ran the actor too)
        bool act_handed = false;
        if constexpr (PERS && ACT) {
            act_handed = chunk_ > 0;
            if (act_handed) act0 = R.actor_actions;      // ([n_envs][CAP], like a pool entry: LOAD adds the intersection's offset)
        }
        if constexpr (ACT) if (!act_handed) {
            // the first tick's actions = actor(rows in HBM), before the state is loaded (nothing else is live): the controlled
            // slots are compacted into `adsts` as if they were dense threads (wave 1's ranks follow wave 0's count)
            if (!PERS || !aprm_staged) {
                const float *gp = (const float *)(R.actor_packed + AP_PRM);
                for (int n = t0_; n < PV_TOTAL; n += CAP) aprm[n] = gp[n];
            }
            const int mt = P.i32[I_META][(size_t)env0_ * CAP + t0_];
            const bool cc = (mt & (M_ALIVE | M_CONTROL)) == (M_ALIVE | M_CONTROL);
            vote<CAP / 64>(sh.m_ctl, t0_, cc);
            adsts[t0_] = 255;                            // (entries beyond the controlled count: no vehicle)
            lds_barrier();
            if (cc) adsts[mask_rank<CAP / 64>(sh.m_ctl, t0_)] = (uint8_t)t0_;
            const int nc = mask_count<CAP / 64>(sh.m_ctl);
            lds_barrier();                               // parameters staged, list complete
            rollout_actor<CAP>(R.actor_packed, aprm, sh.act_next, adsts, nc, R.actor_obs, P.obs_f32 != 0, (size_t)env0_ * CAP, t0_);
            lds_barrier();
        }
        if constexpr (PERS && ACT) {
            if (!aprm_staged && act_handed) {            // (a workgroup whose first item is a later chunk: parameters for its ticks)
                const float *gp = (const float *)(R.actor_packed + AP_PRM);
                for (int n = t0_; n < PV_TOTAL; n += CAP) aprm[n] = gp[n];
            }
            aprm_staged = true;
        }
        if constexpr (PERS) {
            T::template ph_load<true, ACT>(c, P, env0_, t0_, sh, r, act0, true);
            if (Q_TRACE(R) && t0_ == 0) Q_TRACE(R)[((size_t)chunk_ * P.n_envs + env0_) * 8 + 3] = wall_clock64();
        } else T::ph_load(c, P, env0_, t0_, sh, r);     // P.actions = the first tick's actions
        if constexpr (ACT) { if (!act_handed) r.act = sh.act_next[t0_]; }   // (uncontrolled slots: whatever is there, masked in S1)
        if constexpr (IDT) {                            // the first tick's action of the vehicle in this slot: table[row][id]
            r.act = r.alive ? table_action(R.pool, pool_idx, R.table_ids, r.id) : 0.0;
        }
        T::ph_home_store(t0_, sh, r);                   // (HOME: the carried fields to their LDS homes; ordered by the loop's first barrier)
    }
    double sp_act = 0;                                  // IDT: the action of the vehicle this lane spawns
    for (int k = 0; k < n_ticks; k++) {
        // The loop body is one tick of the single-tick kernel.  Without the two opaque copies below the compiler's
        // loop-invariant code motion hoists every kernel-argument load and every per-thread address out of the loop
        // (~150 scalar + ~140 vector registers live across the whole tick -> spills / 2 waves per SIMD).
        // (the SAME variables are re-defined every iteration: a loop-carried value in one register, not an invariant plus a copy)
        // (PERS: the item's intersection travels through the outer loop's phi, which the compiler keeps in a vector register)
        if constexpr (PERS) env0_ = __builtin_amdgcn_readfirstlane(env0_);
        asm volatile("" : "+s"(kav_), "+v"(t0_), "+s"(env0_));
        const KernargPtr ka = kav_;
        const int t = t0_, env = env0_;
#ifndef PVE_NO_RANGE_ASSUME                        // (A/B build knob)
        // the opaque copy hides the range of t: with it the index arithmetic stays in 24 / 32 bits
#ifdef PVE_ASSUME_ONE_WAVE_ONLY                    // (A/B build knob)
        if constexpr (CAP == 64)
#endif
        __builtin_assume(t >= 0 && t < CAP);
#endif
        const PVE_AS4 Const &c = *(const PVE_AS4 Const *)ka;
        const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(ka + OFF_P);
        const PVE_AS4 RolloutArgs &R = *(const PVE_AS4 RolloutArgs *)(ka + OFF_R);
        lds_barrier();
        PVE_PHASE_MARK(0)
        if constexpr (CAP == 64 && !PERS) {
            // One wave per intersection, every intersection of the launch resident at once: the launch lasts as long as its
            // SLOWEST intersection (the one with the most vehicles: +-15 % over the 20 ticks of a short call), and nothing can be
            // re-balanced between workgroups.  So the waves of the crowded intersections get the issue slots first (uniform, one
            // scalar instruction per tick): they run at nearly their solo speed, the light ones -- which finish early anyway --
            // yield.  Thresholds: quartiles of the population at BASELINE config 2's load (31 vehicles on average).
            const int na = __builtin_amdgcn_readfirstlane(sh.hd.n_alive);
#ifndef PVE_NO_LOAD_PRIO                          // (A/B build knob)
            if (na > 38) __builtin_amdgcn_s_setprio(3);
            else if (na > 32) __builtin_amdgcn_s_setprio(2);
            else if (na > 26) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#endif
        }
        if (k > 0) T::ph_tick_init(c, t, sh, r);
        T::ph_step1(c, P, env, t, sh, r);
        lds_barrier();
        PVE_PHASE_MARK(1)
        T::ph_step2(c, t, sh, r);
        T::ph_lists_a(c, t, sh);
        lds_barrier();
        PVE_PHASE_MARK(2)
        const bool last_tick = k + 1 == n_ticks;      // (uniform)
        T::ph_step3(c, t, sh, r, last_tick);
        T::ph_step3_publish(t, sh, r);
        T::ph_lists_b(t, sh);
        lds_barrier();
        PVE_PHASE_MARK(3)
        if constexpr (ShT::HOME) {
            // the 304-entry pool: BUILD .. WALK in passes over groups of lists that fit it (one pass unless the intersection
            // holds ~100 controlled vehicles; pve_tick_core.h, group_end)
            T::ph_build_prep(c, t, sh, r);
            T::ph_scan_init(r);
            if (__builtin_amdgcn_readfirstlane((int)sh.loff[NL]) <= ShT::POOL) {      // (uniform; the rule: every list in ONE pass,
                T::template ph_build_fill<false>(c, t, sh, r, 0, NL);                 //  the very code of the 5 CAP blocks)
                lds_barrier();
                T::template ph_rank<false>(t, sh);
                lds_barrier();
                T::template ph_scan_lists<false>(c, t, sh, r, 0, NL);
            } else
            for (int d0 = 0, pass = 0; d0 < NL; pass++) {
                const int d1 = __builtin_amdgcn_readfirstlane(T::group_end(sh, d0));
                T::template ph_build_fill<true>(c, t, sh, r, d0, d1);
                lds_barrier();
                T::template ph_rank<true>(t, sh, pass, d0, d1);
                lds_barrier();
                T::template ph_scan_lists<true>(c, t, sh, r, d0, d1);
                d0 = d1;
                if (d0 < NL) lds_barrier();           // (the next pass files its entries over the lists this one has just read)
            }
        } else {
        T::ph_build(c, t, sh, r);
        lds_barrier();
        PVE_PHASE_MARK(4)
        T::ph_rank(t, sh);
        lds_barrier();
        PVE_PHASE_MARK(5)
        T::ph_scan(c, t, sh, r);
        }
        PVE_PHASE_MARK(11)
        T::ph_reward(c, t, sh, r);
        lds_barrier();
        PVE_PHASE_MARK(6)
        int nx = -1;
        if (k + 1 < n_ticks) { pool_idx = (pool_idx + 1 == R.n_pool) ? 0 : pool_idx + 1; nx = pool_idx; }
        if constexpr (IDT) {
            int my_id = r.id;
            if constexpr (ShT::HOME) my_id = sh.h_id[t];
            r.act_nx = (nx >= 0 && r.alive) ? table_action(R.pool, nx, R.table_ids, my_id) : 0.0;
        } else T::ph_prefetch_action(P, R, env, t, nx, r);
        T::ph_effects(c, t, sh, r);
        lds_barrier();
        PVE_PHASE_MARK(7)
        T::ph_prefetch_arrival(P, env, t, sh, r, NL);
        bool sp_late = false;                         // IDT: the intersection is full, FIN decides who spawns (uniform, rare)
        if constexpr (IDT) {
            // the first action of the vehicle lane t spawns at the end of this tick: the spawn votes are complete since the
            // barrier behind FX and, unless the intersection is full, every lane that wants to spawn does (ph_final), so its id
            // is known HERE -- the gather flies under LOCK .. FIN instead of sitting between FIN and STAGE
            const unsigned want = (unsigned)(sh.m_spawn[0] & 0xFFFull);
            sp_late = __builtin_popcount(want) > CAP - sh.hd.n_alive;
            sp_act = 0;
            if (t < NL && nx >= 0 && !sp_late && ((want >> t) & 1u))
                sp_act = table_action(R.pool, nx, R.table_ids, sh.hd.id_seq + __builtin_popcount(want & ((1u << t) - 1u)));
        }
        T::ph_lock(c, t, sh, r, last_tick);
        lds_barrier();
        T::ph_lock2(t, sh, r, last_tick);
        T::ph_keep_prefix(t, sh);
        lds_barrier();                                // (also what orders the previous tick's output stores before this tick's)
        PVE_PHASE_MARK(8)
        if constexpr (!ACT && !IDT) { if (!(ShT::HOME && last_tick)) T::ph_park_action(t, sh, r); }   // (HOME, last tick: the cells hold the jerks)
        const Outputs O = T::template tick_outputs<TRAIN>(P, R, k_base_ + k);
        if constexpr (ACT) {
            adst = -1;
            T::template ph_final<true>(c, P, O, env, t, sh, r, fc, true, &adst);
            adsts[t] = (uint8_t)(adst < 0 ? 255 : adst);  // (dense thread t; threads >= n_ctl: 255)
        } else T::template ph_final<true, !TRAIN>(c, P, O, env, t, sh, r, fc, k + 1 == n_ticks || (TRAIN && O.state_pre != nullptr));
        PVE_PHASE_MARK(9)
        if constexpr (IDT) {
            if (fc.new_slot >= 0) sh.act_next[fc.new_slot] = r.act_nx;   // (act_next = xy32: dead since REWARD)
            // (a full intersection: who spawns is known since FIN only; the gather is in flight under barrier A and STAGE)
            if (sp_late) sp_act = (fc.sp_slot >= 0 && nx >= 0) ? table_action(R.pool, nx, R.table_ids, fc.sp_id) : 0.0;
        }
        HomeRegs hr;
        T::ph_home_take(t, sh, r, fc, hr);                 // (HOME; the last reads of the old arrangement)
        if (fc.still) {                               // (uniform) nobody moves: the registers carry over
            T::ph_stage_header(t, sh, fc);
            T::ph_carry_over(t, sh, r, fc);
        } else {
            lds_barrier();                            // A: nobody reads the tick's work arrays any more
            // (uniform; barrier A also orders the obs_pre rows of this tick.  PERS: the stale rows of an item's first tick are what
            //  ANOTHER workgroup's item stored -- block k - 1 of the trajectory --: coherent loads)
            if constexpr (TRAIN) {
                if (O.state_pre) {                       // (uniform) the 7 x 28 states: descriptors, barrier, cooperative write
                    T::ph_state_publish(O, t, sh, r);
                    lds_barrier();
                    T::template ph_state_coop<PERS>(P, O, env, t, sh);
                }
            }
            T::ph_stage(c, t, sh, r, fc);
            T::ph_home_put(t, sh, fc, hr);
            if constexpr (IDT) { if (fc.sp_slot >= 0) sh.act_next[fc.sp_slot] = sp_act; }
            if constexpr (ACT) {
                // next tick's actions: the vehicle lane t spawns gets the action of an all-zero row (ref :380, :420), the
                // controlled vehicles that stay get actor(row)
                if (fc.sp_slot >= 0) sh.act_next[fc.sp_slot] = (double)aprm[PV_A0];
                // (PERS: also on an item's last tick, unless it is the call's last: the next item starts from these actions)
                const bool hand_on = PERS && k + 1 == n_ticks && chunk_ + 1 < R.n_full + R.n_taper;
                if (k + 1 < n_ticks || hand_on)
                    rollout_actor<CAP>(R.actor_packed, aprm, sh.act_next, adsts, fc.n_ctl, O.obs_post, P.obs_f32 != 0,
                                       (size_t)env * CAP, t);
                if (hand_on) {
                    lds_barrier();                    // (every tile's actions are in act_next)
                    R.actor_actions[(size_t)env * CAP + t] = sh.act_next[t];
                }
            }
            lds_barrier();                            // B: the staging area is complete
            if (k + 1 < n_ticks) T::ph_reload(t, sh, r);
        }
        PVE_PHASE_MARK(10)
    }
    {
        const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(ka0_ + OFF_P);
        T::ph_flush(P, env0_, t0_, sh);
        if (PROF && P.phase_cycles && (t0_ & 63) == 0) {
            unsigned long long *row = P.phase_cycles + ((size_t)env0_ * (CAP / 64) + (t0_ >> 6)) * 16;
#pragma unroll
            for (int k = 0; k < 12; k++) row[k] += pc_[k];
            row[12] += clock64() - sclk0_; row[13] += wall_clock64() - wclk0_;
        }
    }
    if constexpr (PERS) {
        const PVE_AS4 RolloutArgs &R = *(const PVE_AS4 RolloutArgs *)(ka0_ + OFF_R);
        const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(ka0_ + OFF_P);
        q_publish(R, P.n_envs, t0_, env0_, chunk_);
    } else break;
  }
    if constexpr (PERS) q_leave(*(const PVE_AS4 RolloutArgs *)(ka0_ + OFF_R), t0_);
}

// General-geometry tick (lane_num 4 / 8): same workgroup-per-intersection structure, phases of
// pve_tick_geo.h (per-route sorted lists: PAIRS -> RANK -> WALK; the membership scan only as the overflow fallback).
// Registers: 5 waves per SIMD (<= 96 VGPR) for CAP = 128 = 10 workgroups per CU.  CAP = 64 is one wave per workgroup and its
// 9.4 KB of LDS admit 16 workgroups per CU = 4 waves per SIMD whatever the registers: 128 VGPR there (the 4-lane variant,
// which carries both the far-conflict table and its replay fallback, spilled 35 registers at 96).  The 4-lane variant takes
// 128 at CAP = 128 too (8 workgroups per CU instead of 10, no scratch; its fast path is k_rollout_geo anyway).
template <int CAP, bool PROF = false, bool FIX4 = false>
__global__ __launch_bounds__(CAP) __attribute__((amdgpu_waves_per_eu((CAP == 64 || FIX4) ? 4 : 5, (CAP == 64 || FIX4) ? 4 : 5))) void k_tick_geo(const GeoConst g_arg, const Params P_arg)
{
    KernargPtr ka0_ = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const PVE_AS4 GeoConst &g = *(const PVE_AS4 GeoConst *)ka0_;
    const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(ka0_ + ((sizeof(GeoConst) + 7) / 8 * 8));
    __shared__ SharedGeo<CAP> sh;
    const int t = threadIdx.x;
    const int env = blockIdx.x;
    Regs r;
    typedef TickGeo<CAP> T;
    typedef Tick<CAP, SharedGeo<CAP>> B;
    unsigned long long pc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev_ = PROF ? wall_clock64() : 0ull;
#undef PVE_PHASE_MARK
#define PVE_PHASE_MARK(idx)                                                              \
    if (PROF) {                                                                          \
        unsigned long long now_ = wall_clock64();                                        \
        pc_[idx] = now_ - tprev_;                                                        \
        tprev_ = now_;                                                                   \
    }
    T::ph_load(g, P, env, t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(0)
    T::ph_step1(g, P, env, t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(1)
    B::ph_step2(g.base, t, sh, r);
    T::ph_order(t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(2)
    B::ph_step3(g.base, t, sh, r);
    B::ph_step3_publish(t, sh, r);
    T::ph_order2(t, sh, r);
    T::ph_pairs_mode(t, sh, P.geo_scan != 0);
    lds_barrier();
    PVE_PHASE_MARK(3)
    if (T::pairs_over(sh, P.geo_scan != 0)) {         // (uniform, rare: the capacity upper bounds overflow the entry pool)
        T::ph_pairs_count(g, t, sh, P.geo_scan != 0);
        lds_barrier();
        T::ph_pairs_exact(t, sh, P.geo_scan != 0);
        lds_barrier();
        T::ph_pairs_apply(t, sh);
        lds_barrier();
    }
    T::ph_pairs_fill(g, t, sh);
    lds_barrier();
    PVE_PHASE_MARK(4)
    T::ph_load_late(P, env, t, sh, r);
    if (FIX4) T::ph_fix_table(g, t, sh, r);      // (behind FILL: its rows are the back entries' indices)
    T::ph_rank(t, sh, env);
    lds_barrier();
    PVE_PHASE_MARK(5)
    T::template ph_scan<FIX4>(g, t, sh, r);
    PVE_PHASE_MARK(11)
    T::ph_reward(g, t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(6)
    T::ph_effects(g, t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(7)
    B::ph_prefetch_arrival(P, env, t, sh, r, g.lane_num);
    B::ph_lock_slot(g.base, t, sh, r);
    if (P.out.state_pre) T::ph_state_order(t, sh, r);   // (uniform)
    lds_barrier();
    B::ph_lock2_slot(t, sh, r);
    lds_barrier();
    PVE_PHASE_MARK(8)
    T::ph_final(g, P, env, t, sh, r);
    PVE_PHASE_MARK(9)
    if (P.out.state_pre) {
        __threadfence_block();
        __syncthreads();
        T::ph_state(P, env, t, sh, r);
    }
    if (PROF && P.phase_cycles && (t & 63) == 0) {
        unsigned long long *row = P.phase_cycles + ((size_t)env * (CAP / 64) + (t >> 6)) * 16;
#pragma unroll
        for (int k = 0; k < 12; k++) row[k] += pc_[k];
    }
}

// pve_step_many for the 4- / 8-lane layouts: the phases of k_tick_geo in a loop, the state resident on the chip between the
// ticks (TickGeo::ph_final<true> / ph_stage / ph_reload: the staging arrays overlay the list storage, which is dead by
// FIN); the lookup tables are copied to LDS once per launch instead of once per tick, only the per-tick outputs and the
// prefetched actions / arrival times touch HBM.  What it buys beyond the bytes: a launch of k_tick_geo lasts as long as its
// slowest intersection (43 us against a mean of 30 for 4 lanes x 64 slots); here a slow tick of one intersection is
// averaged over the ticks of the launch.
// TRAIN: the training outputs (obs_pre, state_pre) per tick of a trajectory roll-out; IDT: PVE_SRC_TABLE (actions by (tick,
// vehicle id), gathered by the vehicle's own thread and parked at its NEW slot in `p[]`, which is free between FIN and the next
// S1) -- as in k_rollout, variants of their own so that the default kernel keeps its register allocation.
// PERS: the persistent work-queue form (k_rollout<.., PERS>: same queue, same hand-off), every source; with the training outputs
// for 8 lanes only (the 4-lane variant <.., true, 4, TRAIN, .., PERS> would spill 32-46 registers: never instantiated, refused by
// launch_rollout_geo).
// ACT: pve_step_many(PVE_SRC_ACTOR) -- the closed loop of main.py:398-441 for lane_num 4 / 8 resident on the chip, as in
// k_rollout<.., ACT>: behind STAGE every wave runs the actor (pve_actor.h: actor_tile32) on its tiles of 32 controlled
// vehicles, whose float32 / float64 rows FIN has just stored to obs_post; the phases here work per slot, so the list of the
// controlled vehicles' NEW slots (`adsts`, by rank among the controlled vehicles) is filed behind FIN; the actions wait in
// `p[]` (free between FIN and the next S1) where RELOAD looks for them, the spawned vehicles get the action of the all-zero
// row.  Every tick is staged (no `still` shortcut): the actor runs every tick.
template <int CAP, bool FIX4 = false, int WPE = 4, bool TRAIN = false, bool IDT = false, bool PERS = false, bool ACT = false>
__global__ __launch_bounds__(CAP) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_rollout_geo(const GeoConst g_arg, const Params P_arg,
                                                                                                const RolloutArgs R_arg)
{
    static_assert(!(TRAIN && IDT), "lane_num 4 / 8: the table source without the training outputs");
    static_assert(!ACT || !IDT, "the closed loop of the geometry kernel: one action source");
    KernargPtr ka0_ = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t OFF_P = (sizeof(GeoConst) + 7) / 8 * 8, OFF_R = OFF_P + (sizeof(Params) + 7) / 8 * 8;
    __shared__ SharedGeo<CAP> sh;
    __shared__ __attribute__((aligned(64))) float aprm[ACT ? PV_TOTAL : 1];
    __shared__ uint8_t adsts[ACT ? CAP : 1];         // NEW slot of the k-th controlled vehicle (255: gone / no such vehicle)
    __shared__ int q_word[4];
    int t0_ = threadIdx.x;
    int env0_ = PERS ? 0 : blockIdx.x;
    int k_base_ = 0, chunk_ = 0;
    bool aprm_staged = false;                        // PERS + ACT: the float parameters are in LDS (once per workgroup)
    KernargPtr kav_ = ka0_;
    Regs r;
    FinCarry fc;
    typedef TickGeo<CAP> T;
    typedef Tick<CAP, SharedGeo<CAP>> B;
    int pool_idx, n_ticks;
    if constexpr (PERS) { if (t0_ == 0) { q_word[2] = -1; q_word[3] = 0; } }
  for (;;) {                                        // PERS: one pass per item; else exactly one pass
    {
        const PVE_AS4 GeoConst &g = *(const PVE_AS4 GeoConst *)ka0_;
        const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(ka0_ + OFF_P);
        const PVE_AS4 RolloutArgs &R = *(const PVE_AS4 RolloutArgs *)(ka0_ + OFF_R);
        pool_idx = R.pool_tick0;
        n_ticks = R.n_ticks;
        const double *act0 = nullptr;
        if constexpr (PERS) {
            int e, ch, kb, nt;
            if (!q_take(R, P.n_envs, t0_, q_word, e, ch, kb, nt)) break;
            env0_ = e; chunk_ = ch; k_base_ = kb; n_ticks = nt;
            if (R.source == 1 || R.source == 3) pool_idx = (R.pool_tick0 + k_base_) % R.n_pool;
            if (R.source == 1) act0 = R.pool + (size_t)pool_idx * (size_t)P.n_envs * CAP;
        }
        // ACT (cf. k_rollout): the first item of an intersection in a call computes its first actions from the caller's rows,
        // every later item takes them from `actor_actions`, where the previous item's last tick left them
        bool act_handed = false;
        if constexpr (PERS && ACT) {
            act_handed = chunk_ > 0;
            if (act_handed) act0 = R.actor_actions;
        }
        if constexpr (ACT) if (!act_handed) {
            if (!PERS || !aprm_staged) {
                const float *gp = (const float *)(R.actor_packed + AP_PRM);
                for (int n = t0_; n < PV_TOTAL; n += CAP) aprm[n] = gp[n];
            }
            const int mt = P.i32[I_META][(size_t)env0_ * CAP + t0_];
            const bool cc = (mt & (M_ALIVE | M_CONTROL)) == (M_ALIVE | M_CONTROL);
            vote<CAP / 64>(sh.m_ctl, t0_, cc);
            adsts[t0_] = 255;
            lds_barrier();
            if (cc) adsts[mask_rank<CAP / 64>(sh.m_ctl, t0_)] = (uint8_t)t0_;
            const int nc = mask_count<CAP / 64>(sh.m_ctl);
            lds_barrier();                               // parameters staged, list complete
            rollout_actor<CAP>(R.actor_packed, aprm, sh.p, adsts, nc, R.actor_obs, P.obs_f32 != 0, (size_t)env0_ * CAP, t0_);
            lds_barrier();                               // (the actions wait in p[], which LOAD does not touch)
        }
        if constexpr (PERS && ACT) {
            if (!aprm_staged && act_handed) {            // (a workgroup whose first item is a later chunk: parameters for its ticks)
                const float *gp = (const float *)(R.actor_packed + AP_PRM);
                for (int n = t0_; n < PV_TOTAL; n += CAP) aprm[n] = gp[n];
            }
            aprm_staged = true;
        }
        if constexpr (PERS) {
            T::template ph_load<true, ACT>(g, P, env0_, t0_, sh, r, act0, true);
            lds_barrier();
            T::template ph_load_late<true>(P, env0_, t0_, sh, r);
        } else {
        T::ph_load(g, P, env0_, t0_, sh, r);            // P.actions = the first tick's actions
        lds_barrier();
        T::ph_load_late(P, env0_, t0_, sh, r);
        }
        if constexpr (ACT) { if (!act_handed) r.act = sh.p[t0_]; }   // (uncontrolled slots: whatever is there, masked in S1)
        if constexpr (IDT) {                            // the first tick's action of the vehicle in this slot: table[row][id]
            const int idc = r.id < 0 ? 0 : (r.id < R.table_ids ? r.id : R.table_ids - 1);
            r.act = r.alive ? R.pool[(size_t)pool_idx * (size_t)R.table_ids + idc] : 0.0;
        }
    }
    double sp_act = 0;                                  // IDT: the action of the vehicle this lane spawns
    for (int k = 0; k < n_ticks; k++) {
        // (the same opaque re-definitions as in k_rollout: nothing derived from the arguments is hoisted out of the loop)
        if constexpr (PERS) env0_ = __builtin_amdgcn_readfirstlane(env0_);
        asm volatile("" : "+s"(kav_), "+v"(t0_), "+s"(env0_));
        const KernargPtr ka = kav_;
        const int t = t0_, env = env0_;
#ifndef PVE_NO_RANGE_ASSUME                        // (A/B build knob)
        __builtin_assume(t >= 0 && t < CAP);          // (the opaque copy hides the range: index arithmetic stays in 24 / 32 bits)
#endif
        const PVE_AS4 GeoConst &g = *(const PVE_AS4 GeoConst *)ka;
        const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(ka + OFF_P);
        const PVE_AS4 RolloutArgs &R = *(const PVE_AS4 RolloutArgs *)(ka + OFF_R);
        lds_barrier();
        if (k > 0) T::ph_tick_init(g, t, sh, r);      // (the list counters S1 adds to were cleared in the previous tick's FX phase)
        T::ph_step1(g, P, env, t, sh, r);
        lds_barrier();
        B::ph_step2(g.base, t, sh, r);
        T::ph_order(t, sh, r);
        lds_barrier();
        B::ph_step3(g.base, t, sh, r);
        B::ph_step3_publish(t, sh, r);
        T::ph_order2(t, sh, r);
        T::ph_pairs_mode(t, sh, P.geo_scan != 0);
        lds_barrier();
        if (T::pairs_over(sh, P.geo_scan != 0)) {     // (uniform, rare)
            T::ph_pairs_count(g, t, sh, P.geo_scan != 0);
            lds_barrier();
            T::ph_pairs_exact(t, sh, P.geo_scan != 0);
            lds_barrier();
            T::ph_pairs_apply(t, sh);
            lds_barrier();
        }
        T::ph_pairs_fill(g, t, sh);
        lds_barrier();
        if (FIX4) T::ph_fix_table(g, t, sh, r);      // (behind FILL: its rows are the back entries' indices)
        T::ph_rank(t, sh);
        lds_barrier();
        T::template ph_scan<FIX4>(g, t, sh, r);
        T::ph_reward(g, t, sh, r);
        lds_barrier();
        int nx = -1;
        if (k + 1 < n_ticks) { pool_idx = (pool_idx + 1 == R.n_pool) ? 0 : pool_idx + 1; nx = pool_idx; }
        if constexpr (IDT) {
            const int idc = r.id < 0 ? 0 : (r.id < R.table_ids ? r.id : R.table_ids - 1);
            r.act_nx = (nx >= 0 && r.alive) ? R.pool[(size_t)nx * (size_t)R.table_ids + idc] : 0.0;
        } else B::ph_prefetch_action(P, R, env, t, nx, r);
        T::ph_effects(g, t, sh, r);
        T::ph_lists_clear(t, sh);
        lds_barrier();
        B::ph_prefetch_arrival(P, env, t, sh, r, g.lane_num);
        B::ph_lock_slot(g.base, t, sh, r);
        const Outputs O = B::template tick_outputs<TRAIN>(P, R, k_base_ + k);
        if (TRAIN && O.state_pre) T::ph_state_order(t, sh, r);    // (uniform; `ord` is staging storage from FIN on)
        lds_barrier();
        B::ph_lock2_slot(t, sh, r);
        lds_barrier();
        T::template ph_final<true>(g, P, O, env, t, sh, r, fc, ACT || k + 1 == n_ticks || (TRAIN && O.state_pre != nullptr));
        if constexpr (ACT) {
            // the k-th controlled vehicle's NEW slot (m_ctl: the votes of S1); entries beyond the controlled count: no vehicle
            if (t >= fc.n_ctl) adsts[t] = 255;
            if (r.alive && r.ctl) adsts[mask_rank<CAP / 64>(sh.m_ctl, t)] = (uint8_t)(fc.new_slot < 0 ? 255 : fc.new_slot);
        }
        if constexpr (IDT) {
            if (!fc.still && fc.new_slot >= 0) sh.p[fc.new_slot] = r.act_nx;      // (p[] is free between FIN and the next S1)
            sp_act = 0;                               // the vehicle this lane spawns (id known since FIN): its first action
            if (fc.sp_slot >= 0 && nx >= 0) {
                const int idc = fc.sp_id < R.table_ids ? fc.sp_id : R.table_ids - 1;
                sp_act = R.pool[(size_t)nx * (size_t)R.table_ids + idc];
            }
        }
        if (fc.still) {                               // (uniform) nobody moves: the registers carry over
            T::ph_carry_over(t, sh, r, fc);
        } else {
            lds_barrier();                            // A: nobody reads the tick's work arrays any more
            // (uniform; barrier A also orders the obs_pre rows of this tick.  PERS: the stale rows of an item's first tick are what
            //  ANOTHER workgroup's item stored: coherent loads)
            if (TRAIN && O.state_pre) T::template ph_state<PERS>(P, O, env, t, sh, r);
            T::ph_stage(g, t, sh, r, fc);
            if constexpr (IDT) { if (fc.sp_slot >= 0) sh.p[fc.sp_slot] = sp_act; }
            if constexpr (ACT) {
                // next tick's actions: the vehicle lane t spawns gets the action of an all-zero row (ref :380, :420), the
                // controlled vehicles that stay get actor(row)
                if (fc.sp_slot >= 0) sh.p[fc.sp_slot] = (double)aprm[PV_A0];
                // (PERS: also on an item's last tick, unless it is the call's last: the next item starts from these actions)
                const bool hand_on = PERS && k + 1 == n_ticks && chunk_ + 1 < R.n_full + R.n_taper;
                if (k + 1 < n_ticks || hand_on)
                    rollout_actor<CAP>(R.actor_packed, aprm, sh.p, adsts, fc.n_ctl, O.obs_post, P.obs_f32 != 0, (size_t)env * CAP, t);
                if (hand_on) {
                    lds_barrier();                    // (every tile's actions are in p[])
                    R.actor_actions[(size_t)env * CAP + t] = sh.p[t];
                }
            }
            lds_barrier();                            // B: the staging area is complete
            if (k + 1 < n_ticks) {
                T::ph_reload(t, sh, r);
                if constexpr (IDT || ACT) r.act = sh.p[t];
            }
        }
    }
    {
        const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(ka0_ + OFF_P);
        B::ph_flush(P, env0_, t0_, sh);
    }
    if constexpr (PERS) {
        const PVE_AS4 RolloutArgs &R = *(const PVE_AS4 RolloutArgs *)(ka0_ + OFF_R);
        const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(ka0_ + OFF_P);
        q_publish(R, P.n_envs, t0_, env0_, chunk_);
    } else break;
  }
    if constexpr (PERS) q_leave(*(const PVE_AS4 RolloutArgs *)(ka0_ + OFF_R), t0_);
}

// tools/probe_kernel.sh: compile ONE variant of k_rollout / k_rollout_geo (seconds instead of minutes for the whole library) to
// read its register allocation / ISA while working on it: -DPVE_PROBE_ONE="128, 5, false, false, false, true, true" or
// -DPVE_PROBE_GEO="128, true, 4, false, false, true, true"
#if defined(PVE_PROBE_ONE) || defined(PVE_PROBE_GEO)
#ifdef PVE_PROBE_ONE
template __global__ void k_rollout<PVE_PROBE_ONE>(const Const, const Params, const RolloutArgs);
#endif
#ifdef PVE_PROBE_GEO
template __global__ void k_rollout_geo<PVE_PROBE_GEO>(const GeoConst, const Params, const RolloutArgs);
#endif
#else
template <int CAP>
__global__ __launch_bounds__(64) void k_reset_geo(const GeoConst g_arg, const Params P_arg, int cap_ticks)
{
    KernargPtr kar_ = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const PVE_AS4 GeoConst &g = *(const PVE_AS4 GeoConst *)kar_;
    const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(kar_ + ((sizeof(GeoConst) + 7) / 8 * 8));
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env < P.n_envs) reset_env_geo<CAP>(g, P, env, cap_ticks);
}

template <int CAP>
__global__ __launch_bounds__(CAP) void k_compact(const Params P_arg)
{
    const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    __shared__ Shared<CAP> sh;
    const int t = threadIdx.x;
    const int env = blockIdx.x;
    CRegs r;
    Tick<CAP>::ph_c_load(P, env, t, sh, r);
    __syncthreads();
    Tick<CAP>::ph_c_store(P, env, t, sh, r);
}

// diagnostics (pve_debug_traffic_probe): exactly the load pattern of the tick's load phase (6 f64 + 6 i32
// SoA rows per env, 8 B / 4 B per lane, every slot) and one dword store per workgroup -- a kernel with a KNOWN
// byte count to calibrate rocprofv3's FETCH_SIZE for this access width (MI355X_MICROARCH.md, HBM section)
template <int CAP>
__global__ __launch_bounds__(CAP) void k_probe(const Params P, int *sink)
{
    const size_t g = (size_t)blockIdx.x * CAP + threadIdx.x;
    double a = P.f64[F_P][g] + P.f64[F_V][g] + P.f64[F_A][g] + P.f64[F_JERK_SUM][g] + P.f64[F_VIR_DIS][g] +
               P.f64[F_CLOSER_P][g];
    int b = P.i32[I_ID][g] + P.i32[I_SEQ][g] + P.i32[I_VNUM][g] + P.i32[I_STEP][g] + P.i32[I_COUNT][g] +
            P.i32[I_META][g];
    unsigned long long m = __ballot(a + (double)b == 12345.678);
    if (threadIdx.x == 0) sink[blockIdx.x] = (int)m;
}

template <int CAP>
__global__ __launch_bounds__(64) void k_reset(const Const c_arg, const Params P_arg, int cap_ticks)
{
    KernargPtr kar_ = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const PVE_AS4 Const &c = *(const PVE_AS4 Const *)kar_;
    const PVE_AS4 Params &P = *(const PVE_AS4 Params *)(kar_ + ((sizeof(Const) + 7) / 8 * 8));
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env < P.n_envs) reset_env<CAP>(c, P, env, cap_ticks);
}

static std::string hip_err(const char *what, hipError_t e)
{
    return std::string(what) + ": " + hipGetErrorString(e);
}

// hipOccupancyMaxActiveBlocksPerMultiprocessor + the LDS allocation granule.  The runtime's answer divides the CU's 160 KB by
// the block size, but the hardware hands LDS out in 128 granules of 1 280 B per CU (measured, tools/occupancy_probe.hip: ten
// 128-thread workgroups of 15 360 B start at once, of 15 488 .. 16 384 B only nine although the query says ten; sixteen
// one-wave workgroups of 10 240 B = 8 granules do).  A persistent grid sized by the query alone would park its surplus
// workgroups in the dispatcher until the resident ones leave.
template <typename K> static hipError_t resident_blocks(int *nb, K kernel, int threads)
{
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(nb, kernel, threads, 0);
    if (e != hipSuccess) return e;
    hipFuncAttributes fa;
    e = hipFuncGetAttributes(&fa, (const void *)kernel);
    if (e != hipSuccess) return e;
    if (fa.sharedSizeBytes > 0) {
        const int granules = (int)((fa.sharedSizeBytes + 1279) / 1280), by_lds = 128 / granules;
        if (by_lds > 0 && by_lds < *nb) *nb = by_lds;
    }
    return hipSuccess;
}

// Resident workgroups of the persistent roll-out kernels, per DEVICE (a process may hold handles on several GPUs, and they
// need not be the same part): [device][kernel family (12-lane / general geometry)][capacity 64 / 128] workgroups per CU + the
// CU count, filled by the first persistent launch on that device.
struct OccCache {
    static constexpr int MAX_DEV = 64;
    std::mutex mu;
    int wgs[MAX_DEV][4][2] = {};       // family: 0 = k_rollout, 1 = k_rollout_geo, 2 = k_rollout_geo with the actor, 3 = k_rollout<128, 5, ..> (HOME)
    int n_cu[MAX_DEV] = {};
    // -> workgroups the device holds at once for (family, cap), or -1 (err set); `query` = the occupancy call of the variant
    template <typename Q>
    long long resident(int family, int cap, Q query, std::string &err)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { err = "hipGetDevice failed"; return -1; }
        const int ci = cap == 64 ? 0 : 1;
        std::lock_guard<std::mutex> lock(mu);
        const bool cached = dev >= 0 && dev < MAX_DEV && wgs[dev][family][ci] > 0;
        int nb = cached ? wgs[dev][family][ci] : 0, cus = cached ? n_cu[dev] : 0;
        if (!cached) {
            hipDeviceProp_t prop;
            const hipError_t e = query(&nb);
            if (e != hipSuccess || nb <= 0 || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
                err = std::string("occupancy query of the persistent roll-out: ") + hipGetErrorString(e);
                return -1;
            }
            cus = prop.multiProcessorCount;
            if (dev >= 0 && dev < MAX_DEV) { wgs[dev][family][ci] = nb; n_cu[dev] = cus; }
        }
        return (long long)nb * cus;
    }
};
static OccCache g_occ;

struct Backend {
    static int set_device(int dev, std::string &err)
    {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) { err = "no HIP device visible (libpveenv.so needs an AMD GPU; there is no CPU fallback)"; return -1; }
        if (dev < 0 || dev >= n) { err = "device_id out of range"; return -1; }
        return 0;
    }
    static int enter_device(int dev)
    {
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        return prev;
    }
    static void leave_device(int prev)
    {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
    static void *dmalloc(size_t n)
    {
        void *p = nullptr;
        if (hipMalloc(&p, n) != hipSuccess) return nullptr;
        return p;
    }
    static void dfree(void *p) { (void)hipFree(p); }
    static int memset0(void *p, size_t n, void *stream)
    {
        return hipMemsetAsync(p, 0, n, (hipStream_t)stream) == hipSuccess ? 0 : -1;
    }
    // XCDs of the device: the persistent roll-out shards the intersections over them (every shard is worked by one XCD)
    static int n_xcc(int dev)
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || v <= 0) v = 8;
        return v > QUEUE_MAX_SHARDS ? QUEUE_MAX_SHARDS : v;
    }
    static int d2h(void *dst, const void *src, size_t n, void *stream)
    {
        hipStream_t s = (hipStream_t)stream;
        if (hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        return 0;
    }
    static int sync(void *stream, std::string &err)
    {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) { err = hip_err("hipStreamSynchronize", e); return -1; }
        return 0;
    }
    static int check_launch(std::string &err)
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = hip_err("kernel launch", e); return -1; }
        return 0;
    }
    static int launch_tick(const Const &c, const Params &P, int cap, void *stream, std::string &err)
    {
        hipStream_t s = (hipStream_t)stream;
        if (cap == 64) hipLaunchKernelGGL(k_tick<64>, dim3(P.n_envs), dim3(64), 0, s, c, P);
        else hipLaunchKernelGGL(k_tick<128>, dim3(P.n_envs), dim3(128), 0, s, c, P);
        return check_launch(err);
    }
    static int launch_rollout(const Const &c, const Params &P_in, const RolloutArgs &R, int cap, void *stream, std::string &err)
    {
        static const bool off = PVE_KNOB("PVE_NO_ROLLOUT_KERNEL") != nullptr;   // A/B knob: one launch per tick instead
        static const bool act_off = PVE_KNOB("PVE_NO_ROLLOUT_ACTOR") != nullptr;   // A/B knob: actor + tick launches instead
        if (off || (R.source == 2 /* PVE_SRC_ACTOR */ && (act_off || R.exact_f32 || P_in.phase_cycles))) return 1;
        hipStream_t s = (hipStream_t)stream;
        Params P = P_in;
        RolloutArgs Rk = R;
        if (R.source == 1) {
            Rk.pool_tick0 = R.pool_tick0 % R.n_pool;
            P.actions = R.pool + (size_t)Rk.pool_tick0 * (size_t)P.n_envs * (size_t)cap;
        } else P.actions = nullptr;
        if (R.source == 3) Rk.pool_tick0 = R.pool_tick0 % R.n_pool;
        const bool train = P.out.obs_pre || P.out.state_pre;
        const bool act = R.source == 2, idt = R.source == 3, pers = R.queue != nullptr;
        long long grid = P.n_envs;
        if (pers) {                                                           // persistent form (pve_rollout.persistent)
            if (act && (act_off || R.exact_f32 || !R.actor_actions)) return 1;
#ifdef PVE_QUEUE_TRACE
            Rk.q_trace = P.phase_cycles;                                      // (per-item timestamps: make trace)
#else
            if (P.phase_cycles) return 1;       // pve_debug_phase_cycles armed: the chunked launches record the cycles (as the geo path does)
#endif
            // as many workgroups as the chip holds at once (the queue needs no more; fewer when the call has fewer items)
            grid = g_occ.resident(0, cap, [&](int *nb) {
                return (cap == 64) ? resident_blocks(nb, k_rollout<64, 4, false, false, false, false, true>, 64)
                                   : resident_blocks(nb, k_rollout<128, 4, false, false, false, false, true>, 128); }, err);
            if (grid < 0) return -1;
            const long long items = (long long)P.n_envs * (R.n_full + R.n_taper);
            if (const char *g = PVE_KNOB("PVE_PERSISTENT_GRID")) { const long long v = atoll(g); if (v > 0) grid = v; }   // A/B knob
            if (grid > items) grid = items;
        } else if (P.phase_cycles) {
            // the phase-cycle diagnostics exist for the default kernel only: every other variant (table, actor, training outputs)
            // answers "no resident kernel" and the caller falls back to per-tick launches, which record the cycles
            if (train || idt || act) return 1;
            if (cap == 64) hipLaunchKernelGGL((k_rollout<64, 4, true>), dim3(P.n_envs), dim3(64), 0, s, c, P, Rk);
            else hipLaunchKernelGGL((k_rollout<128, 4, true>), dim3(P.n_envs), dim3(128), 0, s, c, P, Rk);
            return check_launch(err);
        }
        // k_rollout<128, 5, ..>: the HOME build of the persistent kernel (128 slots; table, pool and zero sources) -- 96 registers
        // + the 15 264 B block with the carried per-slot fields in LDS homes = 10 workgroups per CU instead of 8, no scratch
        // (pve_tick_core.h: Homes).  Its packed seq_in_lane | id_info[1] word holds arrival cursors < 2^23.  Only the queue form:
        // a plain launch of 4096 workgroups would run as 2560 + 1536.
        // (The training outputs through this block -- k_rollout<128, 5, .., TRAIN, .., PERS>: 94-96 registers -- were measured: 137-141
        //  against 129-131 us per tick; the trainer's roll-out keeps the register build.)
        const bool home_ok = pers && !act && !train && cap == 128 && P.rows < (1 << 23);
        bool home5 = PVE_HOME_DEFAULT != 0 && home_ok;
#ifdef PVE_AB_KNOBS
        if (const char *k5 = getenv("PVE_ROLLOUT_WPE5")) home5 = atoi(k5) != 0 && home_ok;   // A/B knob
#endif
        if (home5) {
            long long g5 = g_occ.resident(3, cap, [&](int *nb) {
                return resident_blocks(nb, k_rollout<128, 5, false, false, false, true, true>, 128); }, err);
            if (g5 < 0) return -1;
            const long long items = (long long)P.n_envs * (R.n_full + R.n_taper);
            if (const char *g = PVE_KNOB("PVE_PERSISTENT_GRID")) { const long long v = atoll(g); if (v > 0) g5 = v; }   // A/B knob
            if (g5 > items) g5 = items;
            if (idt) hipLaunchKernelGGL((k_rollout<128, 5, false, false, false, true, true>), dim3((unsigned)g5), dim3(128), 0, s, c, P, Rk);
            else hipLaunchKernelGGL((k_rollout<128, 5, false, false, false, false, true>), dim3((unsigned)g5), dim3(128), 0, s, c, P, Rk);
            return check_launch(err);
        }
        // variant = capacity x action source (pool / zero, actor, id-indexed table) x training outputs x launch form
#define PVE_ROLLOUT(ACT_, TRAIN_, IDT_, PERS_)                                                                                           \
        do {                                                                                                                             \
            if (cap == 64) hipLaunchKernelGGL((k_rollout<64, 4, false, ACT_, TRAIN_, IDT_, PERS_>), dim3((unsigned)grid), dim3(64), 0, s, c, P, Rk);  \
            else hipLaunchKernelGGL((k_rollout<128, 4, false, ACT_, TRAIN_, IDT_, PERS_>), dim3((unsigned)grid), dim3(128), 0, s, c, P, Rk);          \
        } while (0)
#define PVE_ROLLOUT_SRC(TRAIN_, PERS_)                                                                                                   \
        do {                                                                                                                             \
            if (act) PVE_ROLLOUT(true, TRAIN_, false, PERS_);                                                                            \
            else if (idt) PVE_ROLLOUT(false, TRAIN_, true, PERS_);                                                                       \
            else PVE_ROLLOUT(false, TRAIN_, false, PERS_);                                                                               \
        } while (0)
        if (train) { if (pers) PVE_ROLLOUT_SRC(true, true); else PVE_ROLLOUT_SRC(true, false); }
        else { if (pers) PVE_ROLLOUT_SRC(false, true); else PVE_ROLLOUT_SRC(false, false); }
#undef PVE_ROLLOUT_SRC
#undef PVE_ROLLOUT
        return check_launch(err);
    }
    static int launch_rollout_geo(const GeoConst &g, const Params &P_in, const RolloutArgs &R, int cap, void *stream, std::string &err)
    {
        static const bool off = PVE_KNOB("PVE_NO_ROLLOUT_KERNEL") != nullptr;   // A/B knob: one launch per tick instead
        static const bool act_off = PVE_KNOB("PVE_NO_ROLLOUT_ACTOR") != nullptr;   // A/B knob: actor + tick launches instead
        const bool train = P_in.out.obs_pre || P_in.out.state_pre;
        const bool act = R.source == 2 /* PVE_SRC_ACTOR */;
        if (off || P_in.phase_cycles || (train && R.source == 3)) return 1;
        if (act && (act_off || R.exact_f32 || (R.queue && !R.actor_actions))) return 1;
        const bool fix4 = g.lane_num == 4;           // (the 4-lane layout's far-conflict path is a kernel of its own)
        // (the closed loop with the training outputs is bound by the state writes: its queue form measured 130 against 128 us per
        //  tick of chunked launches, so it stays on those.  Round 6: the 4-lane <TRAIN, PERS> variant is instantiated -- it
        //  carries 32 spilled registers through the tick and is still 6 % faster than chunked launches: 95.3 vs 101.5 us)
        if (R.queue && train && act) return 1;
        hipStream_t s = (hipStream_t)stream;
        Params P = P_in;
        RolloutArgs Rk = R;
        if (R.source == 1) {
            Rk.pool_tick0 = R.pool_tick0 % R.n_pool;
            P.actions = R.pool + (size_t)Rk.pool_tick0 * (size_t)P.n_envs * (size_t)cap;
        } else P.actions = nullptr;
        if (R.source == 3) Rk.pool_tick0 = R.pool_tick0 % R.n_pool;
        dim3 grid(P.n_envs);
        if (R.queue) {
            // the persistent form: as many workgroups as the chip holds at once (the variants of one capacity share their register
            // budget; the actor's parameters add 2 KB of LDS: a query of its own)
            long long gq = g_occ.resident(act ? 2 : 1, cap, [&](int *nb) {
                if (act) return (cap == 64) ? resident_blocks(nb, k_rollout_geo<64, true, 4, false, false, true, true>, 64)
                                            : resident_blocks(nb, k_rollout_geo<128, true, 4, false, false, true, true>, 128);
                return (cap == 64) ? resident_blocks(nb, k_rollout_geo<64, true, 4, false, false, true>, 64)
                                   : resident_blocks(nb, k_rollout_geo<128, true, 4, false, false, true>, 128); }, err);
            if (gq < 0) return -1;
            const long long items = (long long)P.n_envs * (R.n_full + R.n_taper);
            if (gq > items) gq = items;
            grid = dim3((unsigned)gq);
        }
        // (variant = layout x capacity x {default, training outputs, id-indexed table, actor} x launch form)
#define PVE_LAUNCH_GEO_V(CAP_, FIX_, TRAIN_, IDT_, PERS_, ACT_) \
        hipLaunchKernelGGL((k_rollout_geo<CAP_, FIX_, 4, TRAIN_, IDT_, PERS_, ACT_>), grid, dim3(CAP_), 0, s, g, P, Rk)
#define PVE_LAUNCH_GEO(CAP_, FIX_)                                                                                              \
        do {                                                                                                                    \
            if (R.queue) {                                                                                                       \
                if (act) PVE_LAUNCH_GEO_V(CAP_, FIX_, false, false, true, true);                                            \
                else if (train) PVE_LAUNCH_GEO_V(CAP_, FIX_, true, false, true, false);                                          \
                else if (R.source == 3) PVE_LAUNCH_GEO_V(CAP_, FIX_, false, true, true, false);                                   \
                else PVE_LAUNCH_GEO_V(CAP_, FIX_, false, false, true, false);                                                    \
            }                                                                                                                    \
            else if (act && train) PVE_LAUNCH_GEO_V(CAP_, FIX_, true, false, false, true);   /* (round 6: closed loop + training outputs) */ \
            else if (act) PVE_LAUNCH_GEO_V(CAP_, FIX_, false, false, false, true);                                               \
            else if (train) PVE_LAUNCH_GEO_V(CAP_, FIX_, true, false, false, false);                                             \
            else if (R.source == 3) PVE_LAUNCH_GEO_V(CAP_, FIX_, false, true, false, false);                                     \
            else PVE_LAUNCH_GEO_V(CAP_, FIX_, false, false, false, false);                                                       \
        } while (0)
        if (fix4) {
            if (cap == 64) PVE_LAUNCH_GEO(64, true); else PVE_LAUNCH_GEO(128, true);
        } else if (cap == 64) PVE_LAUNCH_GEO(64, false);
        else {
#ifdef PVE_AB_KNOBS
            static const bool w5 = getenv("PVE_ROLLOUT_GEO_WPE5") != nullptr;     // A/B knob: 96-VGPR build, 10 workgroups per CU (414 spills)
            if (w5 && !train && !act && !R.queue && R.source != 3) hipLaunchKernelGGL((k_rollout_geo<128, false, 5>), grid, dim3(128), 0, s, g, P, Rk);
            else
#endif
            PVE_LAUNCH_GEO(128, false);
        }
#undef PVE_LAUNCH_GEO
#undef PVE_LAUNCH_GEO_V
        return check_launch(err);
    }
    static int launch_tick_geo(const GeoConst &g, const Params &P, int cap, void *stream, std::string &err)
    {
        hipStream_t s = (hipStream_t)stream;
        // (the 4-lane layout's far-conflict path is a kernel of its own: FIX4)
        if (g.lane_num == 4) {
            if (P.phase_cycles) {                                    // diagnostics build (pve_debug_phase_cycles)
                if (cap == 64) hipLaunchKernelGGL((k_tick_geo<64, true, true>), dim3(P.n_envs), dim3(64), 0, s, g, P);
                else hipLaunchKernelGGL((k_tick_geo<128, true, true>), dim3(P.n_envs), dim3(128), 0, s, g, P);
            } else if (cap == 64) hipLaunchKernelGGL((k_tick_geo<64, false, true>), dim3(P.n_envs), dim3(64), 0, s, g, P);
            else hipLaunchKernelGGL((k_tick_geo<128, false, true>), dim3(P.n_envs), dim3(128), 0, s, g, P);
        } else if (P.phase_cycles) {
            if (cap == 64) hipLaunchKernelGGL((k_tick_geo<64, true>), dim3(P.n_envs), dim3(64), 0, s, g, P);
            else hipLaunchKernelGGL((k_tick_geo<128, true>), dim3(P.n_envs), dim3(128), 0, s, g, P);
        } else if (cap == 64) hipLaunchKernelGGL((k_tick_geo<64, false>), dim3(P.n_envs), dim3(64), 0, s, g, P);
        else hipLaunchKernelGGL((k_tick_geo<128, false>), dim3(P.n_envs), dim3(128), 0, s, g, P);
        return check_launch(err);
    }
    static int launch_reset_geo(const GeoConst &g, const Params &P, int cap, void *stream, std::string &err)
    {
        hipStream_t s = (hipStream_t)stream;
        const int blocks = (P.n_envs + 63) / 64;
        if (cap == 64) hipLaunchKernelGGL(k_reset_geo<64>, dim3(blocks), dim3(64), 0, s, g, P, 200000);
        else hipLaunchKernelGGL(k_reset_geo<128>, dim3(blocks), dim3(64), 0, s, g, P, 200000);
        return check_launch(err);
    }
    static int launch_compact(const Params &P, int cap, void *stream, std::string &err)
    {
        hipStream_t s = (hipStream_t)stream;
        if (cap == 64) hipLaunchKernelGGL(k_compact<64>, dim3(P.n_envs), dim3(64), 0, s, P);
        else hipLaunchKernelGGL(k_compact<128>, dim3(P.n_envs), dim3(128), 0, s, P);
        return check_launch(err);
    }
    static int pack_actor(const float *W, float *flat, unsigned char *packed, void *stream, std::string &err)
    {
        hipStream_t s = (hipStream_t)stream;
        hipError_t e = hipMemcpyAsync(flat, W, sizeof(float) * AW_TOTAL, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { err = hip_err("hipMemcpyAsync", e); return -1; }
        hipLaunchKernelGGL(k_actor_pack, dim3(1), dim3(256), 0, s, W, packed);
        return check_launch(err);
    }
    template <typename OBS_T>
    static void launch_actor_t(const float *W, const unsigned char *packed, const OBS_T *obs, const int32_t *meta, double *actions,
                               int n_envs, int cap, bool exact_f32, hipStream_t s)
    {
        // persistent workgroups of 4 waves (the parameters are staged in LDS once per workgroup): 4 per CU, one wave
        // per intersection at a time
        static const int wgs = [] { const char *g = PVE_KNOB("PVE_ACTOR_GRID"); const int v = g ? atoi(g) : 0; return v > 0 ? v : 1024; }();
        const int grid = (n_envs + 3) / 4 < wgs ? (n_envs + 3) / 4 : wgs;
        if (exact_f32) {
            if (cap == 64) hipLaunchKernelGGL((k_actor_t<64, OBS_T>), dim3(grid), dim3(256), 0, s, W, obs, meta, actions, n_envs);
            else hipLaunchKernelGGL((k_actor_t<128, OBS_T>), dim3(grid), dim3(256), 0, s, W, obs, meta, actions, n_envs);
        } else {
            if (cap == 64) hipLaunchKernelGGL((k_actor_h<64, OBS_T>), dim3(grid), dim3(256), 0, s, packed, obs, meta, actions, n_envs);
            else hipLaunchKernelGGL((k_actor_h<128, OBS_T>), dim3(grid), dim3(256), 0, s, packed, obs, meta, actions, n_envs);
        }
    }
    static int launch_actor(const float *W, const unsigned char *packed, const void *obs, int mode, const int32_t *meta,
                            double *actions, int n_envs, int cap, void *stream, std::string &err)
    {
        hipStream_t s = (hipStream_t)stream;
        if (mode & 1) launch_actor_t<float>(W, packed, (const float *)obs, meta, actions, n_envs, cap, (mode & 2) != 0, s);
        else launch_actor_t<double>(W, packed, (const double *)obs, meta, actions, n_envs, cap, (mode & 2) != 0, s);
        return check_launch(err);
    }
    static int launch_probe(const Params &P, int cap, int *sink, void *stream, std::string &err)
    {
        hipStream_t s = (hipStream_t)stream;
        if (cap == 64) hipLaunchKernelGGL(k_probe<64>, dim3(P.n_envs), dim3(64), 0, s, P, sink);
        else hipLaunchKernelGGL(k_probe<128>, dim3(P.n_envs), dim3(128), 0, s, P, sink);
        return check_launch(err);
    }
    static int launch_reset(const Const &c, const Params &P, int cap, void *stream, std::string &err)
    {
        hipStream_t s = (hipStream_t)stream;
        const int blocks = (P.n_envs + 63) / 64;
        if (cap == 64) hipLaunchKernelGGL(k_reset<64>, dim3(blocks), dim3(64), 0, s, c, P, 200000);
        else hipLaunchKernelGGL(k_reset<128>, dim3(blocks), dim3(128 / 2), 0, s, c, P, 200000);
        return check_launch(err);
    }
};

#include "pve_capi.inc"
#endif   // PVE_PROBE_ONE / PVE_PROBE_GEO
