"""The FETCH_SIZE / WRITE_SIZE factors every traffic summary applies (tools/leg_traffic_summary.py, tools/pmc_summary.py)."""
import glob, json, os, re


def calibrated_factors():
    """(fetch factor, write factor, source): what one counted byte stands for, from the newest profiles/rNN_counter_calibration.json
    (tools/prof_calibration.sh: launches of the step kernels that move a known byte count)"""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*_counter_calibration.json")),
                   key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)), reverse=True)
    for f in files:
        try:
            fa = json.load(open(f))["factors_applied"]
            return float(fa["FETCH_SIZE"]), float(fa["WRITE_SIZE"]), os.path.relpath(f, here)
        except Exception:
            continue
    return 2.0, 1.0, "MI355X_MICROARCH.md (uncalibrated for this access pattern)"
