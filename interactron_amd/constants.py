"""Tables the evaluators need (reference utils/constants.py).

``ACTIONS``: the four navigation actions, index = policy output (utils/constants.py:4).
``THOR_CLASS_IDS``: indices of the 112 AI2-THOR object classes inside the reference's 1235-entry LVIS-style class table
(utils/constants.py:173 computes it from two name lists; only the resulting integers matter to the metric, so they
are stored directly -- derived in the build container by importing the reference's module).
"""
ACTIONS = ["MoveAhead", "MoveBack", "RotateLeft", "RotateRight"]

NUM_CLASSES = 1235   # index 1235 = "no object"

THOR_CLASS_IDS = [
    3, 11, 18, 22, 57, 67, 76, 126, 131, 132, 138, 142, 149, 180, 193, 218, 229, 231, 272, 283, 284, 343, 360, 366, 389,
    394, 414, 420, 429, 467, 468, 476, 535, 603, 614, 621, 625, 630, 640, 686, 693, 707, 718, 737, 747, 756, 780, 781,
    789, 803, 817, 834, 835, 837, 880, 909, 926, 954, 956, 960, 978, 981, 992, 998, 999, 1007, 1050, 1070, 1076, 1078,
    1094, 1096, 1097, 1098, 1107, 1108, 1138, 1160, 1170, 1187, 1203, 1204, 1205, 1206, 1207, 1208, 1209, 1210, 1211,
    1212, 1213, 1214, 1215, 1216, 1217, 1218, 1219, 1220, 1221, 1222, 1223, 1224, 1225, 1226, 1227, 1228, 1229, 1230,
    1231, 1232, 1233, 1234,
]
