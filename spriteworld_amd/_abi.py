"""ctypes mirror of include/swb.h (struct layouts and enums of the C ABI)."""
import ctypes as C

SWB_MAX_SPRITES = 16
SWB_MAX_TASKS = 8
SWB_MAX_SHAPES = 32
SWB_MAX_SHAPE_VERTS = 64
SWB_MAX_GROUPS = 8
SWB_MAX_CUTS = 4
SWB_MAX_CELLS = (SWB_MAX_CUTS + 1) * (SWB_MAX_CUTS + 1)
SWB_MAX_CANDIDATES = 12

# enum swb_action_space
ACTION_SELECT_MOVE, ACTION_DRAG_AND_DROP, ACTION_EMBODIED = 0, 1, 2
# enum swb_task_kind
TASK_NO_REWARD, TASK_FIND_GOAL, TASK_CLUSTERING = 0, 1, 2
# enum swb_meta_aggregator / swb_meta_termination
AGG_SUM, AGG_MAX, AGG_MIN, AGG_MEAN = 0, 1, 2, 3
TERM_ALL, TERM_ANY = 0, 1
# enum swb_step_type (== dm_env.StepType)
STEP_FIRST, STEP_MID, STEP_LAST = 0, 1, 2
# enum swb_env_error
ENV_ERR_DB_ZERO, ENV_ERR_DB_LABELS, ENV_ERR_SPAN_OVERFLOW = 1, 2, 4
# enum swb_sprite_attr
ATTR_SHAPE, ATTR_ANGLE, ATTR_SCALE = 0, 1, 2


class SwbTask(C.Structure):
  _fields_ = [
      ('kind', C.c_int32),
      ('sparse_reward', C.c_int32),
      ('goal_position', C.c_double * 2),
      ('weights_dimensions', C.c_double * 2),
      ('terminate_distance', C.c_double),
      ('raw_reward_multiplier', C.c_double),
      ('terminate_bonus', C.c_double),
      ('termination_threshold', C.c_double),
      ('reward_range', C.c_double),
      ('n_xcuts', C.c_int32),
      ('n_ycuts', C.c_int32),
      ('xcuts', C.c_double * SWB_MAX_CUTS),
      ('ycuts', C.c_double * SWB_MAX_CUTS),
  ]


class SwbConfig(C.Structure):
  _fields_ = [
      ('n_envs', C.c_int32),
      ('max_sprites', C.c_int32),
      ('image_h', C.c_int32),
      ('image_w', C.c_int32),
      ('anti_aliasing', C.c_int32),
      ('bg_rgb', C.c_uint8 * 4),
      ('action_space', C.c_int32),
      ('action_scale', C.c_double),
      ('motion_cost', C.c_double),
      ('keep_in_frame', C.c_int32),
      ('max_episode_length', C.c_int32),
      ('pos_is_f32', C.c_int32),
      ('n_tasks', C.c_int32),
      ('is_meta', C.c_int32),
      ('meta_aggregator', C.c_int32),
      ('meta_termination', C.c_int32),
      ('action_is_f32', C.c_int32),
      ('meta_terminate_bonus', C.c_double),
      ('tasks', SwbTask * SWB_MAX_TASKS),
  ]


class SwbPool(C.Structure):
  _fields_ = [
      ('n_entries', C.c_int32),
      ('n_sprites', C.c_void_p),
      ('x', C.c_void_p),
      ('y', C.c_void_p),
      ('x_vel', C.c_void_p),
      ('y_vel', C.c_void_p),
      ('scale', C.c_void_p),
      ('cos_a', C.c_void_p),
      ('sin_a', C.c_void_p),
      ('shape', C.c_void_p),
      ('rgb', C.c_void_p),
      ('label', C.c_void_p),
      ('cell_label', C.c_void_p),
      ('pool_base', C.c_void_p),
      ('pool_len', C.c_void_p),
      ('angle', C.c_void_p),
      ('color', C.c_void_p),
      ('attr_f32', C.c_void_p),
  ]


class SwbOutputs(C.Structure):
  _fields_ = [
      ('obs', C.c_void_p),
      ('reward', C.c_void_p),
      ('discount', C.c_void_p),
      ('step_type', C.c_void_p),
      ('success', C.c_void_p),
      ('error', C.c_void_p),
  ]


class SwbState(C.Structure):
  _fields_ = [
      ('x', C.c_void_p),
      ('y', C.c_void_p),
      ('n_sprites', C.c_void_p),
      ('pool_entry', C.c_void_p),
      ('step_count', C.c_void_p),
      ('reset_next', C.c_void_p),
      ('episode', C.c_void_p),
  ]


class SwbVariantInfo(C.Structure):
  _fields_ = [('nw', C.c_int32), ('ncol', C.c_int32), ('vs', C.c_int32), ('lds_bytes_per_wave', C.c_int32),
              ('waves_per_simd', C.c_int32), ('resample_waves_per_simd', C.c_int32), ('n_bands', C.c_int32),
              ('n_column_groups', C.c_int32), ('run_cap', C.c_int32), ('paint_in_cover', C.c_int32),
              ('arena_units', C.c_int32), ('reserved_', C.c_int32), ('run_list_bytes', C.c_int64)]


FACTOR_UNIFORM_F32, FACTOR_UNIFORM_INT, FACTOR_DISCRETE = 0, 1, 2


class SwbFactor(C.Structure):
  _fields_ = [
      ('kind', C.c_int32),
      ('n', C.c_int32),
      ('lo', C.c_double),
      ('hi', C.c_double),
      ('cand', C.c_double * SWB_MAX_CANDIDATES),
  ]


FACTOR_ORDER = ('x', 'y', 'scale', 'angle', 'c0', 'c1', 'c2', 'x_vel', 'y_vel')   # enum swb_factor_index
SWB_N_FACTORS = len(FACTOR_ORDER)
SWB_MAX_HOLDOUTS = 2


class SwbHoldout(C.Structure):
  _fields_ = [
      ('redraw_mask', C.c_uint32),
      ('box_mask', C.c_uint32),
      ('lo', C.c_double * SWB_N_FACTORS),
      ('hi', C.c_double * SWB_N_FACTORS),
  ]


class SwbSpriteGroup(C.Structure):

  def factor(self, key):
    return self.factors[FACTOR_ORDER.index(key)]

  _fields_ = [
      ('count_min', C.c_int32),
      ('count_max', C.c_int32),
      ('factors', SwbFactor * SWB_N_FACTORS),
      ('n_holdouts', C.c_int32),
      ('reserved', C.c_int32),
      ('holdouts', SwbHoldout * SWB_MAX_HOLDOUTS),
      ('n_shapes', C.c_int32),
      ('shapes', C.c_int32 * SWB_MAX_CANDIDATES),
      ('cos_a', C.c_double * SWB_MAX_CANDIDATES),
      ('sin_a', C.c_double * SWB_MAX_CANDIDATES),
      ('label', C.c_int8 * SWB_MAX_TASKS),
  ]


SWB_MAX_ALTERNATIVES = 16


class SwbAlternative(C.Structure):
  _fields_ = [
      ('n', C.c_int32),
      ('group', C.c_int32 * SWB_MAX_GROUPS),
  ]


class SwbSampler(C.Structure):
  _fields_ = [
      ('n_groups', C.c_int32),
      ('shuffle', C.c_int32),
      ('color_map', C.c_int32),
      ('n_alternatives', C.c_int32),
      ('alternatives', SwbAlternative * SWB_MAX_ALTERNATIVES),
      ('deg_cos', C.c_double * 360),
      ('deg_sin', C.c_double * 360),
      ('groups', SwbSpriteGroup * SWB_MAX_GROUPS),
  ]
