"""The reduced system's factorisation routines alone, hot, on one workgroup (timing-only build -DCC_RIG_TIMING:
scripts/build_variant.sh rigtime cc_rig.hip --patch timing -DCC_RIG_TIMING; CC_LIB_PATH=scripts/ablate_build/libcc_rigtime.so).
which 0: chol_block4 (four columns at a time, all waves), 1: eight-column panels on wave 0 + trailing updates on the matrix pipe."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from camera_calibrator_amd import capi
for S in (18, 30, 42, 51, 63):
    for which in (0, 1, 2):
        if which == 1 and S > 64:
            continue
        out = np.zeros(3)
        for reps in (50,):
            capi._check(capi.lib().cc_rig_debug_chol_bench(C.c_int32(S), C.c_int32(reps), C.c_int32(which), out.ctypes.data_as(C.POINTER(C.c_double))))
            print(json.dumps({"S": S, "routine": ["chol_block4", "panel8 + mfma trailing", "chol_block4 WITHOUT the trailing update of waves 1..3 (timing only)"][which], "reps": reps, "us_per_factorisation": round(out[0] / 100.0, 2),
                              "cycles": round(out[1]), "checksum": out[2] / reps}))
