"""profiles/r02_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd databases) of
`bench.py --serial --no-ba --cpu-sample 0 --no-single --batch B`.

Corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports both counters in KB.  WRITE_SIZE is exact for streaming
stores.  FETCH_SIZE under-reports reads on gfx950: exactly 1/2 for 16-byte-per-lane streams; other widths must be
calibrated on a known byte count in the same access pattern.  These kernels read with 4-byte loads, so the factor is
calibrated on k_blur's read side: a once-through streaming read of every pyramid level (known size: sum of pyramid pixels)
with the same 4-byte loads.  hbm_bytes_per_launch = FETCH_SIZE x factor + WRITE_SIZE.
usage: make_traffic_json.py fetch.db write.db B out.json"""
import json
import sys

sys.path.insert(0, "profiles/probes")
import pmc_summary as P

fetch, write, B, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
f, w = P.load(fetch), P.load(write)
P_FRONT, P_BIRD = 2853088, 811960
mean = lambda v: sum(v) / len(v)
blur_known = (P_FRONT + P_BIRD) * B / 2.0            # mean bytes READ per k_blur launch (front and bird launches alternate)
factor = blur_known / (mean(f[("k_blur", "FETCH_SIZE")]) * 1024.0)
kern = {}
for (k, cn), vals in f.items():
    if not k.startswith("k_") or (k, "WRITE_SIZE") not in w:
        continue
    fr, wr = mean(vals) * 1024.0, mean(w[(k, "WRITE_SIZE")]) * 1024.0
    kern[k] = {"fetch_raw_bytes_per_launch": fr, "write_bytes_per_launch": wr, "fetch_corrected_bytes_per_launch": fr * factor,
               "hbm_bytes_per_launch": fr * factor + wr, "launches": len(vals)}
json.dump({"batch": B, "fetch_correction_factor": factor,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB -> bytes); FETCH_SIZE x %.3f "
                     "(gfx950 under-reports reads; factor calibrated on k_blur's once-through 4-byte-load read of known size) + WRITE_SIZE; mean per launch" % factor,
           "kernels": kern}, open(out, "w"), indent=1)
print("factor", factor)
for k, v in sorted(kern.items()):
    print("%-28s fetch raw %8.1f MB  corrected %8.1f MB  write %8.1f MB  total %8.1f MB" % (k, v["fetch_raw_bytes_per_launch"] / 1e6, v["fetch_corrected_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, v["hbm_bytes_per_launch"] / 1e6))
