"""Fold one GPU visit's raw counter files (gpurun_out/<tag>/, written by tools/gpu_round4.sh) into the tracked summaries bench.py reads:
    python tools/collect_profiles.py r04 <commit the visit ran at>
  profiles/pmc_sq.json        SQ_INSTS_VALU / SQ_WAVES ... per kernel (rocprofv3 --pmc, tools/gpu_pmc_sq.sh), keyed by short names
  profiles/pmc_traffic.json   HBM bytes per launch (FETCH_SIZE x2-corrected + WRITE_SIZE, tools/gpu_pmc.sh): this round's values added
                              under detail[<kernel>]["fetch_<tag>"/"write_<tag>"], the headline kernels' totals refreshed
both stamped ("_stamp") with the commit and the date of the visit, which bench.py prints next to every figure it takes from them."""
import datetime
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, commit = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
stamp = f"collected {datetime.date.today().isoformat()} at commit {commit[:12]}, visit {tag}"

SHORT = {"imu_integrate_multi_kernel": "imu_integrate_multi", "imu_cov_seg_kernel": "imu_cov_seg", "imu_integrate_bwd_kernel": "imu_integrate_bwd",
         "lm_se3inv_trial2_kernel": "lm_se3inv_trial2", "lm_se3inv_finish_kernel": "lm_se3inv_finish",
         "scan_bwd_left_kernel<float, pplie::MulSO3": "scan_bwd_left_so3", "scan_bwd_right_kernel<float, pplie::MulSO3": "scan_bwd_right_so3",
         "scan_bwd_left_kernel<float, pplie::MulSE3": "scan_bwd_left_se3", "scan_bwd_right_kernel<float, pplie::MulSE3": "scan_bwd_right_se3",
         "scan_kernel<float, pplie::MulSO3": "scan_so3", "scan_kernel<float, pplie::MulSE3": "scan_se3"}
sq = {"_stamp": stamp, "_note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY "
      "SQ_WAIT_INST_ANY, one pass, kernel-trace only (tools/gpu_pmc_sq.sh); averages per launch at the bench sizes (4096 x 1024 IMU / "
      "scans, 10^6 InvNet problems).  SQ_INSTS_VALU counts wave64 instructions of the whole launch; *_CYCLES are quad-cycles "
      "summed over waves (MI355X_MICROARCH.md)."}
for f in ("sq_imu", "sq_imu_train", "sq_lm_invnet", "sq_scan_bwd"):
    p = os.path.join(src, f + ".json")
    if not os.path.exists(p):
        continue
    for name, v in json.load(open(p)).items():
        for pat, short in SHORT.items():
            if pat in name and (short not in sq or v.get("launches", 0) > sq[short].get("launches", 0)):
                sq[short] = {k: (round(x, 1) if isinstance(x, float) else x) for k, x in v.items()}
if len(sq) > 2:            # (a visit without SQ passes keeps the file of the last visit that had them -- its stamp says which)
    json.dump(sq, open(os.path.join(ROOT, "profiles", "pmc_sq.json"), "w"), indent=1)

raw = json.load(open(os.path.join(src, "pmc_raw.json")))
tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
tr = json.load(open(tp))
tr["_stamp"] = stamp
det = tr.setdefault("detail", {})
for k, v in raw.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    fetch, write = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024          # KiB; FETCH x2 on gfx950 (MI355X_MICROARCH.md)
    d = det.setdefault(k, {})
    d[f"fetch_{tag}"], d[f"write_{tag}"] = fetch, write
    if k in ("se3_exp_fwd", "se3_log_fwd", "copy16") or k not in tr:
        tr[k] = fetch + write
ALG = {"scan_bwd_left_so3": 4096 * 1025 * 48, "scan_bwd_right_so3": 4096 * 1025 * 48, "scan_bwd_left_se3": 4096 * 1025 * 84,
       "scan_bwd_right_se3": 4096 * 1025 * 84, "imu_integrate_bwd": 4096 * 1024 * 108, "robust_scale_rows": 400_000 * (24 + 288) * 2 - 400_000 * 24}
for k, b in ALG.items():
    if k in det:
        det[k]["algorithmic_bytes"] = b
json.dump(tr, open(tp, "w"), indent=1)

dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in ("bench_final.json", "prof_headline_kernel_stats.csv", "prof_all_kernel_stats.csv", "pmc_raw.json", "smoke.log", "pcg2_100k.json",
          "pcg_iter.json", "imu_train_kernel_stats.csv", "scan_bwd_kernel_stats.csv", "sq_imu.json", "sq_imu_train.json", "sq_lm_invnet.json",
          "sq_scan_bwd.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
log = os.path.join(src, "pytest_gpu_full.log")
if os.path.exists(log):
    with open(log) as fh, open(os.path.join(dst, "pytest_gpu_tail.log"), "w") as out:
        out.writelines(fh.readlines()[-12:])
print("profiles updated:", stamp)
