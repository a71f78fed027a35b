/*
 * pve_env.h -- C ABI of libpveenv.so: the MI355X-native batched unsignalised-intersection
 * environment (drop-in for the hot path of Mingtzge/PVE-MCC_for_unsignalized_intersection).
 *
 * The reference has no FFI / plugin interface: its boundary is the Python object surface of
 * `TrafficInteraction` that main.py touches.  Each entry point below names the
 * reference call it replaces ("ref :N" = traffic_interaction_scene.py line N).  The Python host
 * mirror (pve-mcc_for_unsignalized_intersection_amd/traffic_interaction_scene.py and batched.py)
 * binds exactly these symbols through ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C, no torch / HIP types: device buffers are `void*` device addresses, the stream is a
 *     `void*` holding a hipStream_t (NULL = default stream).
 *   - every function returns PVE_OK (0) or a negative error code; pve_last_error() gives the text.
 *   - all per-vehicle device buffers are caller-allocated, laid out [n_envs][capacity] (row-major),
 *     "slot" = rank of the vehicle in (lane, j) order inside its environment.
 *   - calls on one handle are asynchronous on the handle's stream and must not be issued
 *     concurrently from several host threads.
 */
#ifndef PVE_ENV_H
#define PVE_ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVE_ABI_VERSION 8
#define PVE_LANES 12          /* physical lanes: lane_num = 4, 8 or 12 (arrays are padded to 12) */
#define PVE_MAX_DIRS 16       /* virtual-lane lists (routes): 12 for lane_num 4 / 12, 16 for lane_num 8 (ref :86, :132, :167) */
#define PVE_OBS_WIDTH 28      /* (o_agent_num + 1) * 4, ref :1295 */
#define PVE_NBR 6             /* ref :1324 hard-codes 6 neighbours */
#define PVE_STATE_ROWS 7
#define PVE_N_METRICS 12

enum {
    PVE_OK = 0,
    PVE_ERR_INVALID = -1,     /* bad argument */
    PVE_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime error (text in pve_last_error) */
    PVE_ERR_STATE = -3,       /* call sequence error (e.g. step before set_arrivals/reset) */
    PVE_ERR_NOMEM = -4
};

/* Constructor arguments of the reference, ref :21-23 (`TrafficInteraction.__init__`), plus
 * args.collision_thr (ref :32).  lane_num = 12 runs the optimised kernel (BASELINE metric); lane_num = 4 or 8
 * (ref :66-145) run the general-geometry kernel.  The 3-lane branch is broken upstream. */
#define PVE_CFG_GENERAL_PATH 0x1   /* flags: use the general-geometry kernel for lane_num = 12 too (cross-checks) */
#define PVE_CFG_OBS_F32      0x2   /* flags: pve_outputs.obs_post and the actor's obs input hold float32 [n_envs][cap][28]
                                      (the type the actor consumes, model_agent_maddpg.py:15; SURVEY.md 8d "FP32 observation
                                      output": 268 instead of 380 algorithmic bytes per slot-step).  Fused ticks only:
                                      obs_pre / state_pre follow the row type (every lane_num); pve_compact(obs) is refused. */
#define PVE_CFG_GEO_SCAN     0x4   /* flags (diagnostics): the general-geometry kernel finds list members by scanning every controlled
                                      vehicle (its fallback when an intersection's lists overflow the LDS entry pool) instead of reading
                                      the per-route lists; results are identical (tested) */
#define PVE_CFG_ACTOR_F32    0x8   /* flags: run the actor as an exact float32 FMA chain (v_mfma_f32_16x16x4_f32, the evaluation order
                                      of csrc/pve_actor.h `actor_canonical`) instead of the default split-half form (every operand
                                      x = hi + lo in float16, three v_mfma_f32_32x32x16_f16 per block, float32 accumulation:
                                      ~1e-6 relative per dot product, 5x less matrix-core time) */
typedef struct pve_config {
    double deltaT;          /* 0.1 */
    double vm, vM;          /* 5, 13   (train(): vm = 6, main.py:230) */
    double am, aM;          /* -3, 3 */
    double v0;              /* 10 */
    double lane_cw;         /* 2.5 */
    double dis_ctl;         /* 150 */
    double collision_thr;   /* 2 (main.py:104) */
    int32_t lane_num;       /* 12 (default), 4 or 8 */
    int32_t flags;          /* PVE_CFG_* */
} pve_config;

/* Per-tick outputs.  Every pointer may be NULL (that output is skipped).  `flags` is written for every slot (0 = the
 * slot held no vehicle); the other per-slot outputs are written only for the slots that held a vehicle at tick start
 * (slots < env_out[PVE_EO_N_PRE]) and are unspecified elsewhere.
 * "pre" arrays are indexed by the slot a vehicle had when the tick started (= the `[lane, j]`
 * the reference reports in `ids`, ref :291); "post" arrays by the slot it has after compaction
 * and spawning (= where the next tick's action for it must be written). */
typedef struct pve_outputs {
    double  *obs_post;      /* [n_envs][cap][28]  (float32 rows with PVE_CFG_OBS_F32) row 0 of the state (ref :1336) of the vehicle now in
                               each slot; zeros for vehicles spawned this tick (ref :380,420) */
    double  *obs_pre;       /* [n_envs][cap][28]  same rows, pre-compaction indexing (`re_state[k][0]`); float32 like obs_post with
                               PVE_CFG_OBS_F32 (lane_num 12) */
    double  *state_pre;     /* [n_envs][cap][7][28] full state incl. neighbour rows (ref :1325-1337), same element type;
                               needs obs_prev_post and obs_pre */
    const double *obs_prev_post; /* obs_post buffer written by the previous tick (stale neighbour rows, ref :1332) */
    double  *reward;        /* [n_envs][cap]  pre; 0 for uncontrolled slots (ref :311-320, 346, 357) */
    int32_t *flags;         /* [n_envs][cap]  pre; PVE_F_* bits | collisions_per_veh << 8 (ref :339) */
    int32_t *lanej;         /* [n_envs][cap]  pre; lane << 16 | j  (the `ids` entry, ref :291) */
    int32_t *nbr;           /* [n_envs][cap][6] pre; lane << 16 | j of the 6 nearest, -1 = none (ref :1391-1405);
                               written for controlled vehicles (PVE_F_CTL) only */
    int32_t *new_slot;      /* [n_envs][cap]  pre -> post slot, -1 if deleted this tick (ref :435-444) */
    int32_t *env_out;       /* [n_envs][PVE_ENV_OUT_N] per-env scalars of this tick, see PVE_EO_* */
} pve_outputs;

/* bits of pve_outputs.flags */
#define PVE_F_ALIVE     0x01  /* slot held a vehicle at tick start */
#define PVE_F_CTL       0x02  /* controlled: appears in `ids`, has reward/obs (ref :282) */
#define PVE_F_DONE      0x04  /* veh["Done"] after this tick (ref :347, 351) */
#define PVE_F_DELETED   0x08  /* in delete_veh (exit or collision, ref :348) */
#define PVE_F_FINISHED  0x10  /* passed the box this tick: reward 5, jerks entry (ref :350-359) */
#define PVE_F_LOCK      0x20  /* veh["lock"] set by the dead-lock scan (ref :1482) */
#define PVE_F_INTENT_SHIFT 6  /* bits 6-7: veh["intention"] (ref :382-394); scene_update emits `ids` in
                                 (lane, intention, j) order (ref :233-275), which differs from slot order for lane_num 4 / 8 */

#define PVE_ENV_OUT_N 8
enum { PVE_EO_N_PRE = 0,      /* vehicles alive at tick start */
       PVE_EO_N_CTL,          /* len(ids) */
       PVE_EO_COLLISIONS,     /* `collisions` of the 9-tuple (ref :337) */
       PVE_EO_LOCK,           /* `lock` of the 9-tuple (ref :365-370) */
       PVE_EO_N_DELETED, PVE_EO_N_FINISHED, PVE_EO_N_SPAWNED,
       PVE_EO_N_POST };       /* vehicles alive after the call */

/* metrics vector of pve_get_metrics (sums over all envs of the handle since reset) */
enum { PVE_M_SLOT_STEPS = 0, PVE_M_ALIVE_STEPS, PVE_M_CTL_STEPS, PVE_M_SPAWNED /* id_seq */,
       PVE_M_PASSED, PVE_M_COLLIDED /* main.py:410-412 count */, PVE_M_LOCKS, PVE_M_SUM_REWARD,
       PVE_M_SUM_JERK /* of passed vehicles */, PVE_M_PASSED_STEPS, PVE_M_OVERFLOW, PVE_M_TICKS };

/* Per-vehicle record for the dict view `env.veh_info[lane][ind]` (ref :396-427, live keys only). */
typedef struct pve_vehicle {
    double p, v, a, jerk, jerk_sum, vir_dis, closer_p;
    int32_t lane, j, id /* id_info[0] */, vnum /* id_info[1] */, seq_in_lane;
    int32_t control, finish, done, collision, step, count, lock, lock_a;
    int32_t vir_header[2];
    int32_t intention, route;
} pve_vehicle;

typedef struct pve_env_info {
    double current_time;                 /* ref :223 */
    int32_t n_alive;
    int32_t lane_count[PVE_LANES];       /* veh_num, ref :206 */
    int32_t veh_rec[PVE_LANES];          /* ref :207 */
    int32_t id_seq, passed_veh, passed_veh_step_total;   /* ref :197-198, 212 */
    int32_t head_valid[PVE_MAX_DIRS], head_lane[PVE_MAX_DIRS], head_j[PVE_MAX_DIRS];  /* virtual_lane_4[d][0][1:3], ref :1517 */
    int32_t overflow;                    /* spawns deferred because the env was full */
    int32_t intention_re;                /* ref :42, :387-392 (lane_num 4 / 8) */
} pve_env_info;

typedef struct pve_handle_s *pve_handle;

int pve_abi_version(void);
const char *pve_last_error(void);
void pve_default_config(pve_config *cfg);

/* Bytes of device workspace a handle needs (persistent SoA vehicle state + env headers). */
size_t pve_workspace_bytes(int n_envs, int capacity);

/* Replaces `TrafficInteraction(arrive_time, dis_ctl, args, ...)` object creation (ref :21) for
 * n_envs independent intersections with `capacity` (64 or 128) vehicle slots each.
 * workspace: device buffer of pve_workspace_bytes() bytes, or NULL to let the library allocate. */
int pve_create(const pve_config *cfg, int n_envs, int capacity, int device_id,
               void *workspace, void *stream, pve_handle *out);
int pve_destroy(pve_handle h);
int pve_set_stream(pve_handle h, void *stream);

/* Arrival streams `arrive_time` (ref :195, main.py:388-389): DEVICE buffer of float64
 * [n_envs][rows][lane_num] (env_stride_rows = rows) or one shared [rows][lane_num] stream (env_stride_rows = 0).
 * Times beyond the run must be padded with +inf.  The buffer must outlive the handle's use of it. */
int pve_set_arrivals(pve_handle h, const double *arrivals, int rows, int env_stride_rows);

/* lane_num = 8 only: the reference draws each new vehicle's intention with random.randint(0, 1) after reseeding
 * `random` from OS entropy (ref :381, :390), i.e. irreproducibly; here the draws are an input stream like the
 * arrivals: DEVICE int32 [n_envs][rows][8] (or shared, env_stride_rows = 0), entry [veh_rec[lane]][lane] in {0,1}
 * picks intention[lane][draw] (ref :125-134).  Not set = all zeros.  Must be set before pve_reset. */
int pve_set_intentions(pve_handle h, const int32_t *choice, int rows, int env_stride_rows);

/* Constructor warm-up (ref :196-220): zero all state, then advance each env's clock tick by tick
 * (spawning, ref :378) until it holds at least one vehicle. */
int pve_reset(pve_handle h);

/* FUSED TICK = `for lane, ind: env.step(lane, ind, a)` (ref :1501, main.py:398-406) +
 * `env.scene_update()` (ref :222) + `env.delete_vehicle()` (ref :435) for every env.
 * actions: device float64 [n_envs][capacity], indexed by current (post-compaction) slot;
 * entries of uncontrolled / empty slots are ignored (main.py:401 passes 0). */
int pve_step_all(pve_handle h, const double *actions, const pve_outputs *out);

/* Split protocol for the single-env compatibility class (same kernels, three launches):
 *   pve_scene_update   = all step() calls + scene_update(); vehicles marked Done stay in place
 *   pve_compact        = delete_vehicle() */
int pve_scene_update(pve_handle h, const double *actions, const pve_outputs *out);
int pve_compact(pve_handle h, double *obs_post /* optional: rows are moved with the vehicles */);

/* MADDPG actor inference on the device (reference model_agent_maddpg.py:23-49 `actor_network`, called per
 * vehicle with batch 1 from main.py:36-45, 404): for every controlled vehicle
 *   actions[env][slot] = 3*tanh(Dense1(relu(LN(Dense64(relu(LN(Dense64(LN(obs[env][slot]))))))))   (float32; the two dense
 *   layers on the matrix cores with split-half operands unless PVE_CFG_ACTOR_F32, see pve_config.flags),
 * 0 for every other slot (main.py:401).
 * pve_set_actor installs the policy (the reference restores it once per run, main.py:380-384 `saver.restore`): weights =
 * DEVICE float32[PVE_ACTOR_N_WEIGHTS] in the order
 * LayerNorm{gamma[28],beta[28]}, dense{kernel[28][64],bias[64]}, LayerNorm_1{gamma,beta}[64],
 * dense_1{kernel[64][64],bias[64]}, LayerNorm_2{gamma,beta}[64], dense_2{kernel[64],bias[1]} (TF variable
 * layouts, checkpoint names `agent1actor/...`); they are copied into the handle's workspace (flat, and packed for the
 * matrix cores: centered over the output units, split into half pairs, in operand order), so the caller's buffer may be
 * reused afterwards; call it again after every update of the weights.
 * pve_actor_forward: obs = [n_envs][cap][28] float64 (= obs_post of the previous tick, zeros after reset);
 * actions: [n_envs][cap] float64 out.  `weights` = NULL uses the installed actor; non-NULL = pve_set_actor(weights) first. */
#define PVE_ACTOR_N_WEIGHTS 6393
int pve_set_actor(pve_handle h, const float *weights);
int pve_actor_forward(pve_handle h, const float *weights, const void *obs /* float64, or float32 with PVE_CFG_OBS_F32 */,
                      double *actions);

/* Closed loop on the device, no host round trip: pve_actor_forward(obs_in -> actions) followed by
 * pve_step_all(actions, out) on the same stream (BASELINE config 5).  out->obs_post may be obs_in itself (the tick
 * never reads observations) unless out->state_pre is requested, which needs the previous rows (obs_prev_post). */
int pve_step_all_actor(pve_handle h, const float *weights, const void *obs_in, double *actions,
                       const pve_outputs *out);

/* MANY TICKS, host out of the loop: the reference's episode loop `for i in range(1000): ... step / scene_update /
 * delete_vehicle` (main.py:397-441) with the action source on the device, as ONE call.  Tick k of the call is exactly
 * pve_step_all() with
 *   PVE_SRC_ZERO   actions = 0 (the zero policy of SURVEY.md 8d)
 *   PVE_SRC_POOL   actions = pool[(pool_tick0 + k) % n_pool], pool = DEVICE float64 [n_pool][n_envs][capacity]
 *   PVE_SRC_TABLE  actions[slot] = table[(pool_tick0 + k) % n_pool][min(id of the vehicle in the slot, table_ids - 1)], table =
 *                  DEVICE float64 [n_pool][table_ids] passed in `pool`: a policy that is a function of (tick, vehicle id), e.g.
 *                  the sin tape of SURVEY.md 8c-ii / BASELINE.md 3, `a = sin(0.37 id + 0.05 tick)`; the vehicle's own thread
 *                  gathers its action (lane_num 12, resident kernel only)
 *   PVE_SRC_ACTOR  actions = pve_actor_forward(actor_weights, observation rows the previous tick stored), i.e. the closed
 *                  loop of main.py:398-441 with the actor of model_agent_maddpg.py:23-49; the first tick reads
 *                  `actor_obs` (zeros after pve_reset, ref :380), later ticks the rows written through out->obs_post,
 *                  which is therefore required (and may be the same buffer as actor_obs when trajectory = 0)
 * and the results are bit-identical to n_ticks separate calls (tested).  The ticks run inside one kernel launch with the
 * intersection state resident on the chip between ticks (k_rollout for lane_num 12, k_rollout_geo for lane_num 4 / 8: no
 * state traffic to HBM, only the per-tick outputs) -- PVE_SRC_ACTOR included: the actor runs inside that kernel on the rows
 * the tick has just stored (with PVE_CFG_ACTOR_F32, or together with the training outputs for lane_num 4 / 8: an actor launch
 * + a tick launch per tick, enqueued from C).
 * trajectory = 0: every tick overwrites the `out` buffers (the last tick's outputs remain; metrics accumulate in the
 * handle as usual); trajectory = 1: every non-NULL `out` buffer holds n_ticks consecutive per-tick blocks
 * ([n_ticks][n_envs][cap]...), the roll-out a trainer consumes.  The training outputs (lane_num 12): obs_pre may be
 * requested in either form; state_pre needs trajectory = 1 (tick k reads the stale neighbour rows from block k - 1 of
 * obs_post, tick 0 from out->obs_prev_post = the rows stored before this call), i.e. a MADDPG trainer gets `re_state` and the
 * 7-action vectors (column 2 of the 7 rows, ref :290) of every tick of the roll-out without leaving the resident kernel. */
enum { PVE_SRC_ZERO = 0, PVE_SRC_POOL = 1, PVE_SRC_ACTOR = 2, PVE_SRC_TABLE = 3 };
typedef struct pve_rollout {
    int32_t n_ticks;
    int32_t source;               /* PVE_SRC_* */
    const double *pool;           /* PVE_SRC_POOL */
    int32_t n_pool, pool_tick0;
    const float *actor_weights;   /* PVE_SRC_ACTOR: NULL = the actor installed by pve_set_actor; else DEVICE float32[PVE_ACTOR_N_WEIGHTS], installed first */
    const void *actor_obs;        /* PVE_SRC_ACTOR: rows the first tick's actor reads (float64, or float32 with PVE_CFG_OBS_F32) */
    double *actor_actions;        /* PVE_SRC_ACTOR: DEVICE scratch float64 [n_envs][capacity] (per-tick launches only: the resident
                                     kernel keeps the actions on the chip and does not touch it) */
    int32_t trajectory;
    int32_t table_ids;            /* PVE_SRC_TABLE: columns of the table (vehicle ids beyond it use the last column) */
    int32_t chunk_ticks;          /* 0: all n_ticks in one launch; > 0: launches of at most chunk_ticks ticks each (same results).
                                     Every workgroup of a launch runs its intersection for all the ticks of the launch, so a
                                     launch lasts as long as its slowest intersection: with several handles stepped on their
                                     own streams, shorter launches let the other handles' workgroups fill the slots that the
                                     fast intersections free (bench.py --chunk). */
    int32_t persistent;           /* 1 (with 0 < chunk_ticks < n_ticks, chunk_ticks <= 255): the whole call is ONE launch of as many
                                     workgroups as the chip holds at once; they pull (intersection, chunk) items from a queue in the
                                     handle's workspace, so nothing waits in launch order for the slowest intersection of a chunk
                                     (the reference's episode loop main.py:397-441 has no such boundary either).  Same results as
                                     persistent = 0.  Eligible: lane_num 12 with every source (ZERO / POOL / TABLE / ACTOR unless
                                     PVE_CFG_ACTOR_F32), with or without trajectory = 1, with or without the training outputs
                                     obs_pre / state_pre; lane_num 4 / 8 with every source (TABLE: without the training outputs, as everywhere for these
                                     layouts); the training outputs through the queue with ZERO / POOL (with ACTOR: the resident
                                     kernel in chunked launches -- that roll-out is bound by its state writes).
                                     Anything else is run as chunked launches (pve_debug_last_launch tells which).
                                     PVE_SRC_ACTOR: `actor_actions` is the hand-off buffer between the items of an intersection
                                     (every item's last tick stores the next actions there, the next item reads them): it must
                                     not alias any other buffer of the call.
                                     A persistent call must run to completion ONCE: its hand-off counters live partly in the
                                     handle (host) and partly in the workspace (device), so it must not be captured into a HIP
                                     graph or replayed; after a failed launch / stream error both sides are reset. */
} pve_rollout;
int pve_step_many(pve_handle h, const pve_rollout *ro, const pve_outputs *out);

/* Host read-back (synchronises the stream). */
int pve_read_env(pve_handle h, int env, pve_env_info *out);
int pve_read_vehicles(pve_handle h, int env, pve_vehicle *out, int max_n, int *n_out);
int pve_get_metrics(pve_handle h, double out[PVE_N_METRICS]);

/* Device views of the persistent per-slot state for zero-copy consumers (actor input masks):
 * field = "p","v","a","jerk","jerk_sum","vir_dis","closer_p" (float64 [n_envs][cap]) or
 * "id","seq","vnum","step","count","meta","hdr" (int32 [n_envs][cap]); meta bit0 = control. */
int pve_state_field(pve_handle h, const char *field, void **dev_ptr, int *elem_bytes);
#define PVE_META_CONTROL 0x1
#define PVE_META_FINISH  0x2
#define PVE_META_DONE    0x4
#define PVE_META_LOCK    0x8

int pve_synchronize(pve_handle h);

/* Diagnostics: accumulate per-phase clock ticks of every wave of the tick kernel into a zeroed DEVICE
 * buffer of uint64 [n_envs * capacity/64][16] (column = phase: load, step1, step2, step3, build, rank,
 * walk, effects, lock, final, state); NULL disables. Used by tools/phase_profile.py; no effect on results. */
int pve_debug_phase_cycles(pve_handle h, uint64_t *dev_counters16);

/* Diagnostics: what the last stepping call on this handle launched. */
enum { PVE_LAUNCH_NONE = 0, PVE_LAUNCH_TICK = 1 /* one launch per tick */, PVE_LAUNCH_RESIDENT = 2 /* one k_rollout launch per chunk */,
       PVE_LAUNCH_PERSISTENT = 3 /* ONE launch for the call, items pulled from the work queue */ };
int pve_debug_last_launch(pve_handle h);

/* Diagnostics: the item schedule pve_step_many lays out for a persistent call of n_ticks ticks with items of at most chunk_ticks
 * ticks (the per-episode loop `for i in range(1000)` of main.py:397 cut into queue items): out[0] = ticks per full item, out[1] =
 * full items, out[2] = tapered items behind them, out[3 .. 10] = their lengths, out[11] = items per intersection.  Host only. */
int pve_debug_item_schedule(int n_ticks, int chunk_ticks, int32_t out[12]);

/* Diagnostics: every later pve_step_all / pve_scene_update launch of the 12-lane kernel (k_tick) RETURNS behind phase n
 * (0 load, 1 step1, 2 step2, 3 step3, 4 build, 5 rank, 6 walk + reward, 7 effects, 8 lock) without writing any state, so
 * that hardware counters of truncated launches on one frozen state attribute instructions and LDS conflicts to phases
 * (tools/phase_counters.sh); n < 0 restores the full tick.  The results of truncated launches are meaningless. */
int pve_debug_stop_phase(pve_handle h, int n);

/* Diagnostics: launch a kernel that performs exactly the tick's state-load pattern over every slot
 * (72 B per slot read: 6 x f64 + 6 x i32) and stores one int per env into dev_sink[n_envs]; a known byte
 * count for calibrating rocprofv3 FETCH_SIZE on this access width. */
int pve_debug_traffic_probe(pve_handle h, int32_t *dev_sink);

#ifdef __cplusplus
}
#endif
#endif /* PVE_ENV_H */
