"""Busy fractions of one kernel from the per-counter means of scripts/refresh_profiles_r06.sh's `pmcbusy` section
(profiles/r06/pmc_intr_persist.csv): python scripts/pmc_busy_summary.py profiles/r06/pmc_intr_persist.csv -> one JSON object.
Units as the counters come on gfx950 (checked against each other on this kernel: SQ_WAVE_CYCLES / SQ_WAVES x 4 = the launch's
length in shader clocks, SQ_BUSY_CYCLES / 32 shader engines and GRBM_GUI_ACTIVE / 8 XCDs agree with it): SQ_WAVE_CYCLES, SQ_WAIT_*,
SQ_ACTIVE_INST_* in units of four clocks summed over waves; SQ_VALU_MFMA_BUSY_CYCLES in clocks summed over the 1024 SIMDs (64 per
v_mfma_f64_16x16x4_f64 = 16 x SQ_INSTS_VALU_MFMA_MOPS_F64 / 4)."""
import csv, json, sys
m = {}
for row in csv.DictReader(open(sys.argv[1])):
    m[row["counter"]] = float(row["mean"])
n_simd = 1024
clocks = 4.0 * m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]          # every wave of a persistent launch lives as long as the launch
out = {
    "source": sys.argv[1],
    "launch_shader_clocks": round(clocks),
    "waves": int(m["SQ_WAVES"]),
    "matrix_pipe_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (n_simd * clocks),
    "valu_issue_active_frac": 4.0 * m["SQ_ACTIVE_INST_VALU"] / (n_simd * clocks),
    "wave_time_waiting_any_frac": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
    "wave_time_waiting_to_issue_frac": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
    "valu_instructions_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
    "mfma_f64_instructions": m["SQ_INSTS_VALU_MFMA_MOPS_F64"] / 4.0,
    "lds_instructions_per_wave": m["SQ_INSTS_LDS"] / m["SQ_WAVES"],
}
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}))
