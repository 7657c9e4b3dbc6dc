"""Robot naming tables and pose priors (data mirrored from reference lib/dataset/const.py:58-91,
100-107, 115-247): Panda, Kuka iiwa7 and Baxter."""

LINK_NAMES = {
    "panda": ["panda_link0", "panda_link2", "panda_link3", "panda_link4", "panda_link6", "panda_link7",
              "panda_hand"],
    "kuka": ["iiwa_link_%d" % i for i in range(8)],
    "baxter": ["torso"] + ["%s_%s" % (side, l) for l in ("upper_shoulder", "lower_shoulder", "upper_elbow", "lower_elbow",
                                                          "upper_forearm", "lower_forearm", "wrist", "hand")
                           for side in ("right", "left")],
}

# links whose visual meshes the self-supervised trainer renders, in the order of their mesh files (reference
# lib/utils/urdf_robot.py:209-219; only the Panda has a mesh list there)
MESH_LINKS = {"panda": ["panda_link%d" % i for i in range(8)] + ["panda_hand"]}

JOINT_NAMES = {
    "panda": ["panda_joint%d" % i for i in range(1, 8)] + ["panda_finger_joint1"],
    "kuka": ["iiwa_joint_%d" % i for i in range(1, 8)],
    "baxter": ["head_pan"] + ["%s_%s" % (side, j) for j in ("s0", "s1", "e0", "e1", "w0", "w1", "w2")
                              for side in ("right", "left")],
}

# Baxter keypoints sit at the origins of these joints, expressed in each joint's PARENT link
# (reference lib/utils/urdf_robot.py:57-74)
BAXTER_KEYPOINT_JOINTS = ["torso_t0"] + ["%s_%s" % (side, j) for j in ("s0", "s1", "e0", "e1", "w0", "w1", "w2", "hand")
                                         for side in ("right", "left")]

JOINT_TO_KP = {"panda": [1, 1, 2, 3, 4, 4, 5, 6], "kuka": [1, 2, 3, 4, 5, 6, 7], "baxter": list(range(1, 16))}

PANDA_LIMB_LENGTH = {"link0-link2": 0.3330, "link2-link3": 0.3160, "link3-link4": 0.0825,
                     "link4-link6": 0.39276, "link6-link7": 0.0880, "link7-hand": 0.1070}
KUKA_LIMB_LENGTH = {"link0-link1": 0.1500, "link1-link2": 0.1900, "link2-link3": 0.2100, "link3-link4": 0.1900,
                    "link4-link5": 0.2100, "link5-link6": 0.19946, "link6-link7": 0.10122}
LIMB_LENGTH = {"panda": list(PANDA_LIMB_LENGTH.values()), "kuka": list(KUKA_LIMB_LENGTH.values())}

INITIAL_JOINT_ANGLE = {
    "zero": {r: {n: 0.0 for n in JOINT_NAMES[r]} for r in ("panda", "kuka", "baxter")},
    "mean": {
        "panda": dict(zip(JOINT_NAMES["panda"], [0.0, 0.0, 0.0, -1.52715, 0.0, 1.8675, 0.0, 0.02])),
        "kuka": {n: 0.0 for n in JOINT_NAMES["kuka"]},
        "baxter": dict({n: 0.0 for n in JOINT_NAMES["baxter"]},
                       right_s1=-0.5499999999999999, left_s1=-0.5499999999999999, right_e1=1.284, left_e1=1.284,
                       right_w1=0.2616018366049999, left_w1=0.2616018366049999),
    },
}

JOINT_BOUNDS = {
    "panda": [[-2.9671, 2.9671], [-1.8326, 1.8326], [-2.9671, 2.9671], [-3.1416, 0.0873],
              [-2.9671, 2.9671], [-0.0873, 3.8223], [-2.9671, 2.9671], [0.0000, 0.0400]],
    "kuka": [[-2.9671, 2.9671], [-2.0944, 2.0944]] * 3 + [[-3.0543, 3.0543]],
    "baxter": [[-1.5708, 1.5708], [-1.7017, 1.7017], [-1.7017, 1.7017], [-2.1470, 1.0470], [-2.1470, 1.0470],
               [-3.0542, 3.0542], [-3.0542, 3.0542], [-0.0500, 2.6180], [-0.0500, 2.6180], [-3.0590, 3.0590],
               [-3.0590, 3.0590], [-1.5708, 2.0940], [-1.5708, 2.0940], [-3.0590, 3.0590], [-3.0590, 3.0590]],
}
