// pve_types.h -- internal data layout shared by the HIP kernels, the C-ABI host code and the
// host-side phase emulator used by the CPU tests (tests/emu).  Not part of the public ABI.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PVE_HD __host__ __device__ __forceinline__
#else
#define PVE_HD inline
#endif

namespace pve {

constexpr int NL = 12;        // physical lanes: 12 (fast path), or 4 / 8 padded to 12 (general-geometry path)
constexpr int ND = 16;        // virtual-lane lists = routes: 12 (4- and 12-lane), 16 (8-lane) (ref :86, :132, :167)
constexpr int MAXK = 8;       // longest lane2lane row is 7 (ref :74-87, :107-124), padded
constexpr int OBSW = 28;      // observation row width, ref :1295
constexpr int NNB = 6;        // neighbours, ref :1324

// meta word of a vehicle slot (persistent)
constexpr int M_CONTROL = 0x1, M_FINISH = 0x2, M_DONE = 0x4, M_LOCK = 0x8;
constexpr int M_LOCKA_POS = 0x10, M_LOCKA_NEG = 0x20;   // lock_a = +1 / -1 (ref :1496-1497)
constexpr int M_DEL = 0x40;                             // in delete_veh, awaiting delete_vehicle() (ref :348)
constexpr int M_ALIVE = 0x80;
constexpr int M_COLL_SHIFT = 8, M_COLL_MASK = 0xFFFF;   // veh["collision"] (ref :333-334)
constexpr int M_INT_SHIFT = 24, M_INT_MASK = 0x3;       // veh["intention"] (ref :382-394); 12-lane: lane % 3, not stored
constexpr int M_LANE_SHIFT = 26, M_LANE_MASK = 0xF;     // veh["lane"]: set at the spawn, never changes -- step() reads it from here
                                                        // instead of comparing the slot with the 12 lane starts

// launch modes
constexpr int MODE_FUSED = 0;     // step* + scene_update + delete_vehicle
constexpr int MODE_SCENE = 1;     // step* + scene_update (Done vehicles stay, marked M_DEL)
constexpr int MODE_COMPACT = 2;   // delete_vehicle only

// Geometry / limits, computed once on the host exactly as the reference's constructor does
// (ref :21-45, :148-152, :182-186) and passed by value to the kernels.
struct Const {
    double deltaT, dt2;            // dt2 = pow(deltaT, 2) (ref :1529)
    double vm, vM, am, aM, v0;
    double abs_am, two_abs_am;     // |am|, 2*|am| (ref :1513-1514)
    double inv_abs_am, inv_two_abs_am;   // their correctly rounded reciprocals (div_const: exact division by a constant)
    double aM_minus_am;            // float(aM - am) (ref :319)
    double inv_dt, inv_span;       // 1/deltaT, 1/(aM - am): reward-only terms use a multiplication (float output, ~1 ulp)
    double collision_thr, lock_mean_thr;   // thr, thr + 3 (ref :1471, 1495)
    double exit_p;                 // -dis_ctl + int((12+1)/2)*cw (ref :341-342)
    double cw;
    double inbox[3];               // lane_info[m][1] (ref :149-151)
    double spawn_p[3];             // sum(lane_info[m][0:2]) (ref :395)
    // get_virtual_distance table (ref :733-803): ego movement m (0 left, 1 straight), k-th entry
    // of lane2lane[ego]:  delta = (p1 - A) + B ; chosen iff delta > 0 ; vd = C + delta
    double vdA[2][4], vdB[2][4], vdC[2][4];
    double rot_cos[4], rot_sin[4]; // cos/sin(3.141593/2 * approach) (ref :1251, 1287-1288)
    double arc_k;                  // 3.141593 (ref :1259, 1277)
    int8_t l2l[NL][4];             // lane2lane (ref :153-166), -1 = none
    int8_t l2l_inv[NL][4];         // position of ego inside lane2lane[l2l[ego][k]]
};

// Per-environment header (device memory, one per env).
struct EnvHeader {
    double current_time;           // ref :196, 223
    double sum_reward, sum_jerk;   // metrics
    double next_arr[NL];           // arrive_time[veh_rec[l]][l] (+inf past the stream): cached so that the
                                   // spawn test (ref :379) needs no dependent load on the tick's critical path
    long long alive_steps, ctl_steps, ticks;
    int32_t n_alive;
    int32_t lane_start[NL + 1];    // slot range of each lane; lane counts = veh_num (ref :206)
    int32_t veh_rec[NL];           // arrival cursors (ref :207)
    int32_t id_seq, passed, passed_step_total;   // ref :212, 197, 198
    int32_t head_valid;            // bit d: len(virtual_lane_4[d]) > 0 at its last rebuild (ref :1517)
    int32_t head_lane[ND], head_j[ND];           // virtual_lane_4[d][0][1:3] (stale by design); dwords -> scalar loads
    int32_t collided, locks, overflow;
    int32_t intention_re;          // 4-/8-lane spawn counter (ref :387-388, :392)
};

// Persistent per-slot SoA field indices
enum { F_P = 0, F_V, F_A, F_JERK, F_JERK_SUM, F_VIR_DIS, F_CLOSER_P, NF64 };
enum { I_ID = 0, I_SEQ, I_VNUM, I_STEP, I_COUNT, I_META, I_HDR, NI32 };

// Lookup tables of the general-geometry kernel that are indexed per vehicle / per list: one block, host-computed
// (make_geo_const), copied verbatim into LDS at the top of every launch (a per-lane index into the kernel arguments
// would be a vector load from the argument buffer in the middle of a phase).
struct GeoTab {
    double vd[4][MAXK][4];         // [route type][k] = A, B, C, C2:  delta = (p1 - A) + B ; vd = (delta + C) - C2
    double inbox[4];               // Const::inbox (lane_info[m][1]) of this layout
    int8_t pos[ND][ND];            // pos[d][route] = index of route in lane2lane[d], or -1
    int8_t lst[ND][12];            // lst[r][k] = k-th list (ascending) route r can be filed into, k < nl[r] <= 10
    uint16_t mroutes[ND];          // bit r: route r can be a member of list d (same physical lane, or in lane2lane[d])
    uint16_t lroutes[ND];          // bit d: route r can be filed into list d (transpose of mroutes)
    uint16_t ninv[ND];             // ceil(32768 / nl[r]): exact division of a pair offset < 2048 by nl[r]
    int8_t opp[ND];                // lane2lane[d][1] (4-lane fix-up, ref :1303)
    int8_t dir_lane[ND], dir_index[ND];    // inverse of direction
    int8_t nl[ND];
    int8_t dty[ND];                // d % tmod: row of vd for list d
};

// General geometry (lane_num 4 / 8, and 12 for cross-checks): everything the 4-/8-lane branches of the reference
// derive in the constructor (ref :66-145), get_virtual_distance (ref :453-660) and get_p (ref :896-1249).
struct GeoConst {
    Const base;                    // scalars; inbox / spawn_p / exit_p are those of this lane_num
    int32_t lane_num, dir_num, tmod, RL;   // tmod: routes of the same type are d % tmod apart; RL: left radius / cw
    double H;                      // half box width: 2cw (4-lane), 4cw (8-lane)
    double fix_d, fix_hi, fix_lo;  // 4-lane far-conflict fix-up (ref :1304-1318): (_alpha-alpha)*3*cw, _alpha*3*cw, alpha*3*cw
    int8_t l2l[ND][MAXK];          // lane2lane rows, -1 padded
    int8_t direction[NL][4];       // direction[lane][intention], -1 = none (ref :88-93, :135-144, :168-181)
    int8_t turn[NL];               // get_p: quarter turns of the canonical path (ref :896-1249)
    int8_t pad_[4];
    // the two tables the kernel indexes per vehicle, packed into scalars (a dynamic index into the kernel arguments
    // is a global load): turn[lane] in 4-bit fields, direction[lane][m] + 1 in 5-bit fields of dir_pk[m]
    unsigned long long turn_pk, dir_pk[3];
    GeoTab tab;
};

struct Outputs {        // mirrors pve_outputs (include/pve_env.h)
    double *obs_post, *obs_pre, *state_pre;
    const double *obs_prev_post;
    double *reward;
    int32_t *flags, *lanej, *nbr, *new_slot, *env_out;
};

struct Params {
    EnvHeader *headers;
    double *f64[NF64];           // each [n_envs][cap]
    int32_t *i32[NI32];
    const double *actions;       // [n_envs][cap] or null (all zero)
    const double *arrivals;      // [rows][12] per env
    long long arr_env_stride;    // doubles between envs (0 = shared stream)
    int32_t rows;
    int32_t n_envs;
    int32_t mode;
    int32_t mask_uncontrolled;   // 1: actions of uncontrolled slots are forced to 0 (main.py:401)
    int32_t obs_f32;             // PVE_CFG_OBS_F32: out.obs_post holds float32 rows (the type the actor consumes)
    int32_t geo_scan;            // PVE_CFG_GEO_SCAN: general-geometry kernel uses the membership scan even when the lists fit
    const int32_t *choice;       // 8-lane: the randint(0,1) draws of ref :390, [rows][lane_num] per env, or null (all 0)
    long long choice_env_stride; // int32 elements between envs (0 = shared)
    unsigned long long *phase_cycles;   // diagnostics: 16 counters of wave-cycles per phase, or null
    int32_t stop_phase;          // diagnostics (pve_debug_stop_phase): k_tick returns behind this phase; < 0: the full tick
    Outputs out;
};

// pve_step_many: action source and output addressing of a multi-tick launch (k_rollout)
struct RolloutArgs {
    const double *pool;          // PVE_SRC_POOL: [n_pool][n_envs][cap]
    const unsigned char *actor_packed;   // PVE_SRC_ACTOR: the packed actor parameters (pve_actor.h, k_actor_pack)
    const void *actor_obs;       // PVE_SRC_ACTOR: the observation rows the first tick's actor reads
    const void *prev_rows;       // state_pre: the rows the tick before this launch stored (stale neighbour rows of its first tick)
    int32_t n_ticks, source, n_pool, pool_tick0, trajectory;
    int32_t table_ids;           // PVE_SRC_TABLE: `pool` is [n_pool][table_ids], indexed by (tick, vehicle id)
    int32_t exact_f32;           // PVE_CFG_ACTOR_F32: the resident kernel has no exact-float32 actor (per-tick launches instead)
    // persistent form (pve_rollout.persistent): ONE launch for the whole call; its workgroups pull (intersection, chunk)
    // items from the queue below instead of owning one intersection each (k_rollout<.., PERS>)
    unsigned *queue;             // RolloutQueue words followed by done[n_envs] (device, in the handle's workspace)
    int32_t call_ticks;          // ticks of the whole call (n_ticks = ticks per full item here)
    // item schedule: n_full items of n_ticks ticks, then n_taper shorter ones (taper[k] ticks each): the last items of a call
    // are short so that the chip drains evenly (a workgroup's last item is what it is still busy with when the queue is empty)
    int32_t n_full, n_taper;
    uint8_t taper[8];
    int32_t n_shards;            // env e belongs to shard e % n_shards; a shard is worked by ONE XCD
    uint32_t done_base;          // done[e] - done_base = items of intersection e completed in this call
    double *actor_actions;       // persistent closed loop: [n_envs][cap] actions handed from one item of an intersection to the next
    unsigned long long *q_trace; // diagnostics (pve_debug_phase_cycles armed): [chunk][env][8] timestamps of every item, or null
};

// Item `chunk` of a persistent call -> its first tick within the call and its length (the schedule pve_step_many lays out:
// n_full items of n_ticks ticks, then the taper).  Shared by the kernel's dequeue (q_take) and the test emulator, which runs
// the items sequentially, so the schedule arithmetic of pve_capi.inc is under CPU parity as well.
template <typename RA>
PVE_HD void rollout_item(const RA &R, int chunk, int &k_base, int &n_ticks)
{
    k_base = (chunk < R.n_full ? chunk : R.n_full) * R.n_ticks;
    for (int q = 0; q < chunk - R.n_full; q++) k_base += R.taper[q];                  // (uniform: scalar loop over <= 7 entries)
    n_ticks = chunk < R.n_full ? R.n_ticks : (int)R.taper[chunk - R.n_full];
}

// Work queue of the persistent roll-out (device words, zero between launches: the last workgroup to leave clears them).
// done[e] (behind this block) counts the items of intersection e completed since pve_reset, cumulatively (wrap-safe compares).
constexpr int QUEUE_MAX_SHARDS = 16;
struct RolloutQueue {
    // one 128-byte line per shard: 2048 workgroups pulling from ONE line serialise at ~12 ns per atomic (measured: 15 us
    // mean start-up delay and a congested tail when every exiting workgroup probed the other shards with atomics)
    struct Shard {
        unsigned head;                  // items handed out
        unsigned owner;                 // 0 = nobody yet, else 1 + HW_REG_XCC_ID of the XCD whose workgroups work the shard
        unsigned pad_[30];
    } s[QUEUE_MAX_SHARDS];
    unsigned exits;                     // workgroups that have left the launch
    unsigned pad_[31];
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// pve_set_actor keeps two images of the actor in the handle's workspace: the flat float32 weights (the exact float32
// kernel k_actor_t reads them) and the packed split-half form (pve_actor.h: AP_BYTES_PADDED)
constexpr size_t ACTOR_FLAT_BYTES = 25600, ACTOR_PACKED_BYTES = 26880;

struct Layout {
    size_t off_headers, off_f64[NF64], off_i32[NI32], off_actor_flat, off_actor_packed, off_queue, total;
};

inline Layout make_layout(int n_envs, int cap)
{
    Layout L;
    size_t o = 0;
    L.off_headers = o; o = align_up(o + sizeof(EnvHeader) * (size_t)n_envs, 256);
    for (int k = 0; k < NF64; k++) { L.off_f64[k] = o; o = align_up(o + 8 * (size_t)n_envs * cap, 256); }
    for (int k = 0; k < NI32; k++) { L.off_i32[k] = o; o = align_up(o + 4 * (size_t)n_envs * cap, 256); }
    L.off_actor_flat = o; o = align_up(o + ACTOR_FLAT_BYTES, 256);
    L.off_actor_packed = o; o = align_up(o + ACTOR_PACKED_BYTES, 256);
    L.off_queue = o; o = align_up(o + sizeof(RolloutQueue) + 4 * (size_t)n_envs, 256);
    L.total = o;
    return L;
}

}  // namespace pve
