"""Robot naming tables and pose priors (data mirrored from reference lib/dataset/const.py:58-91,
100-107, 115-222); only the Panda entries are needed by the shipped configs of this build."""

LINK_NAMES = {
    "panda": ["panda_link0", "panda_link2", "panda_link3", "panda_link4", "panda_link6", "panda_link7",
              "panda_hand"],
    "kuka": ["iiwa_link_%d" % i for i in range(8)],
}

JOINT_NAMES = {
    "panda": ["panda_joint%d" % i for i in range(1, 8)] + ["panda_finger_joint1"],
    "kuka": ["iiwa_joint_%d" % i for i in range(1, 8)],
}

JOINT_TO_KP = {"panda": [1, 1, 2, 3, 4, 4, 5, 6], "kuka": [1, 2, 3, 4, 5, 6, 7]}

PANDA_LIMB_LENGTH = {"link0-link2": 0.3330, "link2-link3": 0.3160, "link3-link4": 0.0825,
                     "link4-link6": 0.39276, "link6-link7": 0.0880, "link7-hand": 0.1070}

INITIAL_JOINT_ANGLE = {
    "zero": {"panda": {n: 0.0 for n in JOINT_NAMES["panda"]}, "kuka": {n: 0.0 for n in JOINT_NAMES["kuka"]}},
    "mean": {
        "panda": dict(zip(JOINT_NAMES["panda"], [0.0, 0.0, 0.0, -1.52715, 0.0, 1.8675, 0.0, 0.02])),
        "kuka": {n: 0.0 for n in JOINT_NAMES["kuka"]},
    },
}

JOINT_BOUNDS = {
    "panda": [[-2.9671, 2.9671], [-1.8326, 1.8326], [-2.9671, 2.9671], [-3.1416, 0.0873],
              [-2.9671, 2.9671], [-0.0873, 3.8223], [-2.9671, 2.9671], [0.0000, 0.0400]],
}
