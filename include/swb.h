/*
 * swb.h -- C ABI of the MI355X-native batched Spriteworld step/render engine.
 *
 * The reference (google-deepmind/spriteworld) has no FFI / plugin registry: its
 * hot path is one duck-typed Python call chain.  This
 * header is therefore the boundary a maintainer would bind from Python
 * (ctypes) to replace, for N environments at once:
 *
 *   spriteworld/environment.py:74-78    Environment.reset
 *   spriteworld/environment.py:88-108   Environment.step
 *   spriteworld/environment.py:80-86    Environment.success / should_terminate
 *   spriteworld/action_spaces.py:65-104 SelectMove.step (+ DragAndDrop :133-137)
 *   spriteworld/action_spaces.py:172-214 Embodied.step
 *   spriteworld/tasks.py:70-81,126-158,196-245,248-296  NoReward /
 *                                       FindGoalPosition / Clustering /
 *                                       MetaAggregated reward + success
 *   spriteworld/renderers/pil_renderer.py:67-91  PILRenderer.render
 *   spriteworld/sprite.py:96-138        Sprite geometry, move, contains_point
 *
 * Conventions: every function returns 0 on success and a negative swb_status on
 * error (message via swb_last_error()); no exceptions cross the boundary; plain
 * pointers and sizes only.  "dev" pointers are HIP device pointers owned by the
 * caller (e.g. torch tensors' data_ptr()); "host" pointers are ordinary host
 * memory.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 * One caller thread per handle; all launches are asynchronous w.r.t. `stream`.
 *
 * The same structs (swb_config, swb_task) are consumed by the CPU oracle
 * (oracle/sw_oracle.c) so that tests feed both sides identical inputs.
 */
#ifndef SWB_H_
#define SWB_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWB_MAX_SPRITES 16   /* per environment                                 */
#define SWB_MAX_TASKS 8      /* sub-tasks of a MetaAggregated task              */
#define SWB_MAX_SHAPES 32
#define SWB_MAX_SHAPE_VERTS 64 /* per shape (reference max is 30, the "circle") */
#define SWB_MAX_CUTS 4       /* thresholds per axis of a task filter that keys on position (see swb_task)          */
#define SWB_MAX_CELLS ((SWB_MAX_CUTS + 1) * (SWB_MAX_CUTS + 1))

enum swb_status {
  SWB_OK = 0,
  SWB_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
  SWB_ERR_HIP = -2,       /* HIP runtime error                        */
  SWB_ERR_NO_DEVICE = -3, /* no usable gfx950 device                  */
  SWB_ERR_STATE = -4      /* call order (e.g. step before set_pool)   */
};

/* action_spaces.py: which class the `actions` buffer is interpreted as. */
enum swb_action_space {
  SWB_ACTION_SELECT_MOVE = 0,   /* f64[N,4] (or f32, see action_is_f32)  action_spaces.py:29-111  */
  SWB_ACTION_DRAG_AND_DROP = 1, /* f64[N,4] (or f32)                     action_spaces.py:114-137 */
  SWB_ACTION_EMBODIED = 2       /* i32[N,2]  action_spaces.py:140-221 */
};

enum swb_task_kind {
  SWB_TASK_NO_REWARD = 0,  /* tasks.py:70-81   */
  SWB_TASK_FIND_GOAL = 1,  /* tasks.py:84-158  */
  SWB_TASK_CLUSTERING = 2  /* tasks.py:161-245 */
};

/* tasks.py:250-256 MetaAggregated.REWARD_AGGREGATOR / TERMINATION_CRITERION. */
enum swb_meta_aggregator { SWB_AGG_SUM = 0, SWB_AGG_MAX = 1, SWB_AGG_MIN = 2, SWB_AGG_MEAN = 3 };
enum swb_meta_termination { SWB_TERM_ALL = 0, SWB_TERM_ANY = 1 };

/* dm_env.StepType values (environment.py:78,105-108). */
enum swb_step_type { SWB_STEP_FIRST = 0, SWB_STEP_MID = 1, SWB_STEP_LAST = 2 };

/* Per-environment error bits.  swb_outputs.error is STICKY: a step ORs its bits into the buffer and
 * never clears it, so a caller that polls every k steps misses nothing; the caller zeroes the buffer
 * after reading it.  (Where the reference raises -- Davies-Bouldin ValueError / ZeroDivisionError --
 * the reward of that step is NaN and success is 0.) */
enum swb_env_error {
  SWB_ENV_OK = 0,
  SWB_ENV_ERR_DB_ZERO = 1,       /* Davies-Bouldin score 0 => reference raises ZeroDivisionError (tasks.py:215) */
  SWB_ENV_ERR_DB_LABELS = 2,     /* sklearn check_number_of_labels => reference raises ValueError            */
  SWB_ENV_ERR_SPAN_OVERFLOW = 4  /* internal raster span list overflow (never expected)                      */
};

/* One (sub-)task.  Field names follow the reference attributes they lower
 * (tasks.py:118-124 FindGoalPosition, tasks.py:189-194 Clustering). */
typedef struct swb_task {
  int32_t kind;               /* swb_task_kind                                        */
  int32_t sparse_reward;      /* _sparse_reward                                       */
  double goal_position[2];    /* FindGoal: _goal_position                             */
  double weights_dimensions[2];/* FindGoal: _weights_dimensions                       */
  double terminate_distance;  /* FindGoal: _terminate_distance                        */
  double raw_reward_multiplier;/* FindGoal: _raw_reward_multiplier                    */
  double terminate_bonus;     /* both: _terminate_bonus                               */
  double termination_threshold;/* Clustering: _termination_threshold                  */
  double reward_range;        /* Clustering: _reward_range                            */
  /* Filters / cluster distributions that key on POSITION (factors x, y): the reference evaluates
   * `contains(sprite.factors)` at every step (tasks.py:134-137, 196-205) and a sprite's membership changes as it moves.
   * Every position test of the reference's factor distributions is an interval test `lo <= v < hi`
   * (factor_distributions.py:105-112), so membership is constant on the cells of the grid the tests' bounds cut the plane
   * into: xcuts / ycuts are those bounds (ascending, as values of the position dtype: a bound compared with a float32
   * position is rounded the way numpy compares it), a sprite at (x, y) is in cell
   *   cy * (n_xcuts + 1) + cx,   cx = #{k : x >= xcuts[k]},  cy = #{k : y >= ycuts[k]},
   * and its label this step is swb_pool::cell_label[entry][task][sprite][cell].  n_xcuts = n_ycuts = 0: the task does not
   * key on position and swb_pool::label holds its labels. */
  int32_t n_xcuts, n_ycuts;
  double xcuts[SWB_MAX_CUTS];
  double ycuts[SWB_MAX_CUTS];
} swb_task;

typedef struct swb_config {
  int32_t n_envs;             /* N                                                    */
  int32_t max_sprites;        /* S <= SWB_MAX_SPRITES (per-episode count may be less) */
  int32_t image_h;            /* PILRenderer image_size[0]: a multiple of 4, <= 256   */
  int32_t image_w;            /* PILRenderer image_size[1]                            */
  int32_t anti_aliasing;      /* PILRenderer anti_aliasing (>= 1; canvas width AA*image_size[0] <= 1023) */
  uint8_t bg_rgb[4];          /* PILRenderer bg_color (4th byte unused)               */
  int32_t action_space;       /* swb_action_space                                     */
  double action_scale;        /* SelectMove/DragAndDrop _scale ; Embodied _step_size  */
  double motion_cost;         /* _motion_cost                                         */
  int32_t keep_in_frame;      /* Environment keep_in_frame                            */
  int32_t max_episode_length; /* Environment max_episode_length                       */
  int32_t pos_is_f32;         /* 1: sprite positions are np.float32 arrays (config-   */
                              /* sampled sprites); 0: float64             */
  int32_t n_tasks;            /* 1 for a plain task; >=1 with is_meta                 */
  int32_t is_meta;            /* task is tasks.MetaAggregated                         */
  int32_t meta_aggregator;    /* swb_meta_aggregator                                  */
  int32_t meta_termination;   /* swb_meta_termination                                 */
  int32_t action_is_f32;      /* 1: actions are float32[N,4] (the dtype action_spec()  */
                              /* declares, action_spaces.py:62-63): motion, click point*/
                              /* and cost then follow numpy's float32 arithmetic       */
  double meta_terminate_bonus;/* MetaAggregated _terminate_bonus                      */
  swb_task tasks[SWB_MAX_TASKS];
} swb_config;

/* A pool of initial states ("what init_sprites() returned"), host memory,
 * entry-major: element e*S + s.  Entry e holds n_sprites[e] <= S sprites in
 * back-to-front order (sprite_generators.py / environment.py:75).
 * Everything static per episode is resolved on the host by the caller:
 *   cos_a/sin_a = math.cos/sin(math.radians(angle))     (sprite.py:96-101)
 *   rgb         = renderer _color_to_rgb(sprite.color)  (pil_renderer.py:82)
 *   label[t]    = FindGoal: filter_distrib.contains(factors) (tasks.py:134-137)
 *                 Clustering: first matching cluster index or -1 (tasks.py:196-205)
 *                 (evaluated once per episode: exact for every factor that cannot change inside an episode)
 *   cell_label  for tasks whose filter / cluster distributions key on x or y (swb_task::n_xcuts / n_ycuts): the same label
 *                 for every cell of the task's position grid, evaluated with the sprite's other factors -- the kernel
 *                 looks the sprite's cell up at every step, as the reference re-evaluates contains() at every step.
 *                 (Velocities are constant inside an episode: a filter on x_vel / y_vel is an ordinary label.)
 * Environment n draws entries pool_base[n] + (k mod pool_len[n]), k = 0,1,2...
 */
typedef struct swb_pool {
  int32_t n_entries;          /* P                                                    */
  const int32_t* n_sprites;   /* [P]                                                  */
  const double* x;            /* [P,S]                                                */
  const double* y;            /* [P,S]                                                */
  const double* x_vel;        /* [P,S]                                                */
  const double* y_vel;        /* [P,S]                                                */
  const double* scale;        /* [P,S]                                                */
  const double* cos_a;        /* [P,S]                                                */
  const double* sin_a;        /* [P,S]                                                */
  const int32_t* shape;       /* [P,S]  index into the uploaded shape table           */
  const uint8_t* rgb;         /* [P,S,4] (r,g,b,unused)                               */
  const int8_t* label;        /* [P,n_tasks,S]                                        */
  const int8_t* cell_label;   /* [P,n_tasks,S,SWB_MAX_CELLS]; may be NULL when no task keys on position */
  const int32_t* pool_base;   /* [N]                                                  */
  const int32_t* pool_len;    /* [N]                                                  */
  const double* angle;        /* [P,S]   degrees; only for swb_factors (may be NULL)  */
  const double* color;        /* [P,S,3] c0,c1,c2; only for swb_factors (may be NULL) */
  const uint8_t* attr_f32;    /* [P,S]   bit 0 / bit 1: the sprite's angle / scale is an np.float32 in the reference (what
                               * factor distributions draw; Discrete candidates and constructor arguments are Python numbers),
                               * so that a setter takes `a - self._angle` / `s - self._scale` in that type (sprite.py:163,173
                               * under NEP 50).  May be NULL (= all Python numbers).  swb_sample_pool records it per sprite. */
} swb_pool;

/* Device-side reset sampling: a declarative form of the sprite
 * generators the shipped configs build from Product-of-Continuous/Discrete factor distributions
 * (factor_distributions.py:81-158,268-310; sprite_generators.py:27-70,101-128).  Each group draws
 * `count` sprites (uniform integer in [count_min, count_max]) whose factors are independent; the
 * groups are concatenated in order and the z-order optionally shuffled.  Sampling uses a
 * counter-based Philox4x32-10 stream per pool entry: statistically, not bitwise, equivalent to the
 * reference's global MT19937 draws.  The value TYPES follow the reference exactly: a Continuous
 * factor is np.random.uniform(lo, hi) cast to its dtype (float32 by default, or an integer type),
 * a Discrete factor yields the Python object from its candidate list (a Python float). */
#define SWB_MAX_GROUPS 8
#define SWB_MAX_CANDIDATES 12
enum swb_factor_kind {
  SWB_FACTOR_UNIFORM_F32 = 0, /* Continuous(key, lo, hi)                 -> np.float32     */
  SWB_FACTOR_UNIFORM_INT = 1, /* Continuous(key, lo, hi, dtype='int32')  -> truncated int  */
  SWB_FACTOR_DISCRETE = 2     /* Discrete(key, candidates)               -> Python float   */
};
typedef struct swb_factor {
  int32_t kind;
  int32_t n;                          /* candidates (SWB_FACTOR_DISCRETE)                  */
  double lo, hi;                      /* [lo, hi)   (uniform kinds)                        */
  double cand[SWB_MAX_CANDIDATES];
} swb_factor;
enum swb_factor_index {
  SWB_F_X = 0, SWB_F_Y, SWB_F_SCALE, SWB_F_ANGLE, SWB_F_C0, SWB_F_C1, SWB_F_C2, SWB_F_XVEL, SWB_F_YVEL, SWB_N_FACTORS
};
/* SetMinus(base, hold_out) (factor_distributions.py:313-344): the factors in `redraw_mask` (base.keys)
 * are redrawn until they leave the box  AND_k lo[k] <= value_k < hi[k]  over `box_mask`. */
#define SWB_MAX_HOLDOUTS 2
typedef struct swb_holdout {
  uint32_t redraw_mask, box_mask;     /* bit i = factor index i                            */
  double lo[SWB_N_FACTORS], hi[SWB_N_FACTORS];
} swb_holdout;
typedef struct swb_sprite_group {
  int32_t count_min, count_max;
  /* x, y: UNIFORM_F32 only (float32 positions); angle: DISCRETE, or UNIFORM_INT within [0, 360] */
  swb_factor factors[SWB_N_FACTORS];
  int32_t n_holdouts, reserved;
  swb_holdout holdouts[SWB_MAX_HOLDOUTS];
  int32_t n_shapes, shapes[SWB_MAX_CANDIDATES];                /* shape table indices      */
  double cos_a[SWB_MAX_CANDIDATES], sin_a[SWB_MAX_CANDIDATES]; /* of angle.cand (radians)  */
  int8_t label[SWB_MAX_TASKS];                                 /* task label of the group  */
} swb_sprite_group;
/* sprite_generators.sample_generator (sprite_generators.py:73-98, uniform p): each episode uses ONE
 * alternative, a list of groups (indices into swb_sampler.groups) generated in that order. */
#define SWB_MAX_ALTERNATIVES 16
typedef struct swb_alternative {
  int32_t n;
  int32_t group[SWB_MAX_GROUPS];
} swb_alternative;
typedef struct swb_sampler {
  int32_t n_groups;
  int32_t shuffle;          /* sprite_generators.shuffle over the sprites of the first `shuffle`
                             * generated groups (0: none, >= the number generated: all); later
                             * groups keep their place, e.g. the agent body of
                             * examples/goal_finding_embodied.py:88-93                        */
  int32_t color_map;        /* 0: (c0,c1,c2) are RGB ints; 1: renderers.color_maps.hsv_to_rgb */
  int32_t n_alternatives;   /* 0: every group, in order; else one of alternatives[] per episode */
  swb_alternative alternatives[SWB_MAX_ALTERNATIVES];
  double deg_cos[360], deg_sin[360]; /* math.cos/sin(math.radians(d)) for integer degrees  */
  swb_sprite_group groups[SWB_MAX_GROUPS];
} swb_sampler;

/* Per-step outputs, device memory, caller-owned.  Any pointer may be NULL. */
typedef struct swb_outputs {
  uint8_t* obs;        /* u8 [N, H, W, 3]  PILRenderer.render (pil_renderer.py:67-91)      */
  double* reward;      /* f64[N]  NaN on FIRST steps (dm_env reward=None)                  */
  float* discount;     /* f32[N]  NaN on FIRST, 1 on MID, 0 on LAST                        */
  uint8_t* step_type;  /* u8 [N]  swb_step_type                                            */
  uint8_t* success;    /* u8 [N]  task.success(sprites) (renderers Success, handcrafted.py:115-131) */
  uint8_t* error;      /* u8 [N]  swb_env_error bits                                       */
} swb_outputs;

/* Live state snapshot (host memory, caller-allocated) for parity tests and
 * checkpointing (Environment.state(), environment.py:128-134). */
typedef struct swb_state {
  double* x;            /* [N,S] */
  double* y;            /* [N,S] */
  int32_t* n_sprites;   /* [N]   */
  int32_t* pool_entry;  /* [N]   current episode's pool entry */
  int32_t* step_count;  /* [N]   */
  uint8_t* reset_next;  /* [N]   */
  int32_t* episode;     /* [N]   number of resets so far */
} swb_state;

typedef struct swb_engine* swb_handle;

const char* swb_last_error(void);
int swb_version(void);

/* Lifecycle. `device` is the HIP device ordinal. */
int swb_create(const swb_config* cfg, int device, swb_handle* out);
int swb_destroy(swb_handle h);

/* constants.SHAPES (constants.py:27-40): vertices f64 [total,2], offsets[n_shapes+1]. */
int swb_upload_shapes(swb_handle h, const double* verts, const int32_t* offsets, int32_t n_shapes);

/* PIL ImagingResample 8bpc coefficient tables for one axis (host-computed, see
 * spriteworld_amd/lanczos.py): bounds i32[out,2] (xmin,len), coeffs i32[out,ksize]. */
int swb_upload_resample(swb_handle h, int32_t axis /*0=horizontal,1=vertical*/, int32_t out_size,
                        int32_t ksize, const int32_t* bounds, const int32_t* coeffs);

/* Install the reset pool and mark every environment "reset on next step"
 * (Environment.__init__, environment.py:68-70). */
int swb_set_pool(swb_handle h, const swb_pool* pool);

/* Fills a pool of `n_entries` episodes on the device from `spec` (no host sampling, no upload) and
 * marks every environment "reset on next step", like swb_set_pool.  pool_base/pool_len: i32[N].
 * Entry e draws from the Philox stream (seed, first_entry + e): shards of one job pass their global
 * offset and get the episodes a single process would have drawn for the same entries. */
int swb_sample_pool(swb_handle h, const swb_sampler* spec, int32_t n_entries, const int32_t* pool_base_host,
                    const int32_t* pool_len_host, uint64_t seed, uint64_t first_entry, void* stream);

/* Redraws the pool with the spec and layout of the last swb_sample_pool and a new seed, WITHOUT
 * resetting anything: entries an environment is currently playing are left untouched, so a long
 * run calls this every few episodes and never sees an episode twice.  Asynchronous on `stream`
 * (the same stream as swb_step, or ordered with it). */
int swb_resample_pool(swb_handle h, uint64_t seed, uint64_t first_entry, void* stream);

/* Copies the device pool into caller-allocated HOST arrays laid out like swb_set_pool's input
 * (pool->n_entries must equal the device pool's; angle/color may be NULL).  Synchronous. */
int swb_get_pool(swb_handle h, const swb_pool* pool_host);

/* Environment.reset() for all envs: the next swb_step is a FIRST step. */
int swb_reset_all(swb_handle h, void* stream);

/* Environment.step(): actions_dev is f64[N,4] (f32[N,4] when cfg.action_is_f32) for
 * SelectMove/DragAndDrop, or i32[N,2] for Embodied. */
int swb_step(swb_handle h, const void* actions_dev, const swb_outputs* out, void* stream);

/* observation() only (no state change): obs_dev u8[N,H,W,3]. */
int swb_render(swb_handle h, uint8_t* obs_dev, void* stream);

/* Environment.success() (/root/reference/spriteworld/environment.py:80-81: `task.success(self._sprites)` of the sprites AS
 * THEY ARE -- after sprite setters or swb_set_positions, not as the last step left them): success_dev u8[N].  Evaluates the
 * task (tasks.py:153-158, :239-245, :289-296) in the cover kernel's state phase; no state change, no time step, no frame. */
int swb_evaluate(swb_handle h, uint8_t* success_dev, void* stream);

/* Memory of the hand-off lists (what the cover kernel hands the resample / fill kernel: swb_variant_info::run_list_bytes).
 * A handle starts with a list of max(4, max_sprites + 1) units of 8 bytes per canvas row for every environment and group of 64
 * output columns -- enough for ANY scene of convex sprites, about ten times what the usual scene needs (133 KB per environment
 * for 12 sprites at 128x128 with anti_aliasing 5).  This call, made once the handle has rendered a few typical steps, cuts every
 * list's own part down to 1.25 x the longest list of the last launch and adds a shared arena (half of the parts together) into
 * which a list that outgrows its part moves (a segment twice as large); a scene that finds the arena exhausted too is flagged
 * SWB_ENV_ERR_SPAN_OVERFLOW (never silent).  Results do not change.  Blocking (synchronises `stream`, reallocates); no-op when
 * called again.  swb_set_pool / swb_sample_pool restore the full reservation (the new pool may hold denser scenes).
 * run_cap_out (may be NULL): the units of a list's own part afterwards.  The Python engine calls it after its third
 * rendering step. */
int swb_trim_run_lists(swb_handle h, int32_t* run_cap_out, void* stream);

/* SpriteFactors observation (renderers/handcrafted.py:29-82): factors_dev f64[N,S,10] in
 * sprite.FACTOR_NAMES order (x, y, shape, angle, scale, c0, c1, c2, x_vel, y_vel), `shape` as its
 * constants.ShapeType value (1-based); rows >= n_sprites[env] are zero. */
int swb_factors(swb_handle h, double* factors_dev, void* stream);

/* Scalars of ONE environment (environment.py:74-108 keeps them as attributes): out5 = { sprites in the current
 * episode, pool entry it plays, step count, episodes started, 1 if the next step is a reset }.  Synchronises
 * `stream`; five 4-byte copies whatever the batch size (swb_get_state copies every environment). */
int swb_get_env_state(swb_handle h, int32_t env, int32_t* out5, void* stream);
/* swb_pool::attr_f32 of one sprite of the episode environment `env` is playing (bit 0: angle, bit 1: scale). */
int swb_get_sprite_types(swb_handle h, int32_t env, int32_t sprite, int32_t* flags, void* stream);

/* Blocking state access (synchronises `stream`). */
int swb_get_state(swb_handle h, const swb_state* host_state, void* stream);
int swb_set_positions(swb_handle h, const double* x_host, const double* y_host, void* stream);

/* Sprite attribute setters on a LIVE sprite (sprite.py:152-175; pinned by the reference's
 * tests/sprite_test.py:138-174), for sprite `sprite` of environment `env` in its current episode:
 *   SWB_ATTR_SHAPE  value = index into the uploaded shape table: _shape = s; _reset_centered_path()
 *                   (a fresh path from the new shape and the CURRENT scale and angle, :96-101)
 *   SWB_ATTR_ANGLE  value = degrees: the current centred path is rotated by (value - angle)   (:161-165)
 *   SWB_ATTR_SCALE  value = scale:   the current centred path is scaled by (value - scale) -- the
 *                   DIFFERENCE, as the reference does (:171-175), not the ratio
 * The transforms are incremental, exactly like the reference's (matplotlib Affine2D + transform_path,
 * restated on the host): after a setter the sprite's centred path is no longer a function of its
 * factors, so the library keeps the path itself (device memory, allocated at the first call: N x S x
 * SWB_MAX_SHAPE_VERTS x 16 B) and the engine switches to the kernel build that reads it.  hit-tests
 * (contains_point), rendering and the SpriteFactors observation all see the new attribute; the
 * overrides end with the episode (the next reset draws fresh sprites, environment.py:74-78).
 *   delta  NULL: the difference (value - old) is taken in float64 from the stored factor.  Non-NULL:
 *          the difference the caller computed (e.g. in float32, when the sprite's factor is an
 *          np.float32 and numpy's promotion rules keep `a - self._angle` in float32).
 *   label  NULL: the sprite keeps its task labels.  Non-NULL: i8[n_tasks] new labels (the caller
 *          re-evaluated `filter.contains(sprite.factors)`, tasks.py:134-137,196-205, for filters that
 *          key on the changed attribute).
 * Blocking (synchronises `stream`); SWB_ERR_STATE when the environment has no live episode (never
 * reset, or its episode just ended).  Not part of the step path: call rates of a few per second. */
enum swb_sprite_attr { SWB_ATTR_SHAPE = 0, SWB_ATTR_ANGLE = 1, SWB_ATTR_SCALE = 2 };
int swb_set_sprite_attr(swb_handle h, int32_t env, int32_t sprite, int32_t attr, double value, const double* delta,
                        const int8_t* label, void* stream);

/* The sprite as the engine currently sees it: shape index, angle, scale, and its centred path
 * (Sprite._centered_path.vertices, sprite.py:96-101) -- path_xy f64[SWB_MAX_SHAPE_VERTS][2], n_verts
 * entries written.  Any output pointer may be NULL.  Blocking. */
/* After swb_set_sprite_attr on a handle with tasks that key on position (swb_task::n_xcuts / n_ycuts): the sprite's labels
 * per cell of each task's grid, re-evaluated by the caller with the new attribute -- cells i8[n_tasks][SWB_MAX_CELLS], laid
 * out as swb_pool::cell_label.  Without the call the sprite keeps the per-cell labels of its pool entry.  Blocking. */
int swb_set_sprite_cell_labels(swb_handle h, int32_t env, int32_t sprite, const int8_t* cells, void* stream);

int swb_get_sprite(swb_handle h, int32_t env, int32_t sprite, int32_t* shape, double* angle, double* scale, int32_t* n_verts,
                   double* path_xy, void* stream);

/* The host arithmetic of the setters alone (no device access; for tests against matplotlib):
 *   SWB_ATTR_SHAPE  out = _reset_centered_path of the unit-scale vertices `in_xy` with scale = a, angle = b (degrees)
 *   SWB_ATTR_ANGLE  out = Affine2D().rotate_deg(a - b).transform_path(in)
 *   SWB_ATTR_SCALE  out = Affine2D().scale(a - b).transform_path(in) */
int swb_sprite_path_op(int32_t attr, double a, double b, int32_t n, const double* in_xy, double* out_xy);

/* Which kernels swb_step launches for this handle -- a step is two kernels on the caller's stream: "cover" (state,
 * geometry, coverage -> run lists; swb_cover_kernel<NW>, NW = 32-pixel canvas words per row, one wave per environment)
 * and "resample" (swb_resample_kernel<VS>, VS output rows in flight; swb_fill_kernel when anti_aliasing = 1), one wave
 * per (environment, group of 64 output columns, band of output rows) -- so that measurement code can label a
 * run by what actually ran.  swb_build_id(): content hash of the sources and flags the library was
 * built from (spriteworld_amd/build.py), "unknown" for a hand build. */
typedef struct swb_variant_info {
  int32_t nw;                 /* cover kernel: 32-pixel words per canvas row it is built for */
  int32_t ncol;               /* output columns per lane of the second kernel (always 1: wider images are column groups) */
  int32_t vs;                 /* resample kernel: output rows in flight (0: anti_aliasing = 1, the fill kernel) */
  int32_t lds_bytes_per_wave; /* cover kernel, = per environment */
  int32_t waves_per_simd;     /* register budget the cover kernel was compiled for */
  int32_t resample_waves_per_simd; /* ... the resample / fill kernel */
  int32_t n_bands;            /* bands of output rows: waves of the second kernel per (environment, column group) */
  int32_t n_column_groups;    /* groups of 64 output columns */
  int32_t run_cap;            /* a run list's own part (8-byte units per environment and column group): max(4, max_sprites + 1) per
                               * canvas row until swb_trim_run_lists, then 1.25 x the longest list written; a list that outgrows
                               * it moves to a segment of the shared arena */
  int32_t paint_in_cover;     /* 1: anti_aliasing = 1 and an image of up to 64 columns -- the cover kernel writes the frame
                               * itself and no second kernel is launched */
  int32_t arena_units;        /* units of the arena the run lists of all environments share for their overflow (0 until the first
                               * launch allocates it); a scene that finds it exhausted flags its environment */
  int32_t reserved_;
  int64_t run_list_bytes;     /* device memory of the hand-off lists: fixed parts + arena (0 until the first launch) */
} swb_variant_info;
int swb_variant(swb_handle h, swb_variant_info* out);
const char* swb_build_id(void);

/* Kernel timing: three HIP events recorded on `stream` per swb_step launch while enabled (before the cover kernel, between the
 * two kernels, after the second); swb_step_time_ms returns (total ms, launches) since enable.  A diagnostic: every event is a
 * completion signal between two kernels that would otherwise follow each other directly -- measured on MI355X, a run of
 * back-to-back steps is 6 % slower with it on.  To time a run, bracket it with ONE pair of events of your own. */
int swb_timing_enable(swb_handle h, int32_t enable);
int swb_step_time_ms(swb_handle h, double* total_ms, int64_t* launches);
/* The same interval split at the event between the two kernels of a step: cover (state, geometry,
 * coverage -> run lists) and resample / fill (run lists -> frames). */
int swb_kernel_times_ms(swb_handle h, double* cover_ms, double* resample_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* SWB_H_ */
