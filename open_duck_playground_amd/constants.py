"""Constants of the Open Duck Mini V2 tasks (mirror of reference playground/open_duck_mini_v2/constants.py).

The reference maps task names to MJCF files; the build ships the compiled models
(`assets/<task>.npz`, tools/compile_models.py) and can also compile an MJCF tree at run time
(`Model.from_xml`)."""
from .model import Model, load_task_model

TASKS = ("flat_terrain", "rough_terrain", "flat_terrain_backlash", "rough_terrain_backlash")

FEET_SITES = ["left_foot", "right_foot"]
LEFT_FEET_GEOMS = ["left_foot_bottom_tpu"]
RIGHT_FEET_GEOMS = ["right_foot_bottom_tpu"]
FEET_GEOMS = LEFT_FEET_GEOMS + RIGHT_FEET_GEOMS
HIP_JOINT_NAMES = ["left_hip_yaw", "left_hip_roll", "left_hip_pitch", "right_hip_yaw", "right_hip_roll", "right_hip_pitch"]
KNEE_JOINT_NAMES = ["left_knee", "right_knee"]
JOINTS_ORDER_NO_HEAD = ["left_hip_yaw", "left_hip_roll", "left_hip_pitch", "left_knee", "left_ankle",
                        "right_hip_yaw", "right_hip_roll", "right_hip_pitch", "right_knee", "right_ankle"]
FEET_POS_SENSOR = [f"{site}_pos" for site in FEET_SITES]
ROOT_BODY = "trunk_assembly"
GRAVITY_SENSOR = "upvector"
GLOBAL_LINVEL_SENSOR = "global_linvel"
GLOBAL_ANGVEL_SENSOR = "global_angvel"
LOCAL_LINVEL_SENSOR = "local_linvel"
ACCELEROMETER_SENSOR = "accelerometer"
GYRO_SENSOR = "gyro"


def task_to_model(task_name: str) -> Model:
    """reference constants.task_to_xml (constants.py:28-34): KeyError for unknown tasks.  `rough_terrain`
    has no XML in the reference either (constants.py:23 points at a missing file)."""
    if task_name not in TASKS:
        raise KeyError(task_name)
    return load_task_model(task_name)


class Robot:
    """What a new robot's constants.py / base.py spell out (reference README.md:74-85), read off a compiled model instead: is it the duck, and
    which actuated joints are leg joints (`JOINTS_ORDER_NO_HEAD`: the joints above a foot, in actuator order)."""

    def __init__(self, is_open_duck: bool, joints_order_no_head):
        self.is_open_duck, self.joints_order_no_head = bool(is_open_duck), list(joints_order_no_head)


def robot_of(model: Model) -> Robot:
    import numpy as np
    a = model.a
    act_names = [str(n) for n in a["names_actuator"]]
    if model.nu == 14 and all(j in act_names for j in JOINTS_ORDER_NO_HEAD) and "neck_pitch" in act_names:
        return Robot(True, JOINTS_ORDER_NO_HEAD)
    # leg joints: actuated joints whose body is a foot's body or one of its ancestors (foot = the body of a FEET_SITES site)
    parent = np.asarray(a["body_parentid"]); above = set()
    for site in FEET_SITES:
        b = int(np.asarray(a["site_bodyid"])[model.site_id(site)])
        while b > 0:
            above.add(b); b = int(parent[b])
    jb = np.asarray(a["jnt_bodyid"]); trn = np.asarray(a["actuator_trnid"]).reshape(model.nu, -1)[:, 0]
    jn = [str(n) for n in a["names_jnt"]]
    return Robot(False, [jn[int(j)] for j in trn if int(jb[int(j)]) in above])
