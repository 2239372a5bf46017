"""Soak of the tracking chain against the oracle chain: many short drives with different seeds, image sizes, local-list modes
and bird feature counts; every frame compared (indices, masks, counters bit-exact; pose <= 1e-4).  usage: chain_soak.py SECONDS [SEED] [SHARE_OF_REFERENCE_KEY_FRAME_DRIVES]"""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np, torch
import test_track_chain_gpu as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
ref_share = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
branches = {}
skipped = 0
g = np.random.default_rng(seed0)
t0, runs, frames, bad, worst = time.time(), 0, 0, [], 0.0
while time.time() - t0 < budget:
    seed = int(g.integers(10000, 10_000_000))
    small = g.random() < 0.8
    wh, bwh, fx = ((640, 480), (384, 384), 250.0) if small else ((1280, 720), (512, 512), 500.0)
    B, K = int(g.integers(1, 4)), int(g.integers(3, 8))
    kw = dict(use_lists=bool(g.integers(0, 2)), granular=bool(g.random() < 0.2), contour=bool(g.random() < 0.85),
              pipelined=bool(g.random() < 0.3), bird_nfeatures=int(g.choice([0, 1000, 600])))
    if kw["granular"]:
        kw["pipelined"] = False
    desc = "seed=%d %s B=%d K=%d %s" % (seed, wh, B, K, kw)
    try:
        if g.random() < ref_share:   # drives that mix TrackWithMotionModel, TrackReferenceKeyFrame and the fall-back
            modes = {k: str(g.choice(["motion", "reference", "motion+reference", "bird", "bird_kf"], p=[0.3, 0.25, 0.2, 0.15, 0.1])) for k in range(1, K)}
            rekey = tuple(int(k) for k in range(1, K) if g.random() < 0.3)
            pick = lambda: int(g.integers(0, B)) if g.random() < 0.4 else None
            desc = "seed=%d %s B=%d K=%d modes=%s rekey=%s" % (seed, wh, B, K, modes, rekey)
            few = {int(g.integers(0, B)): int(g.choice([8, 30, 60, 80, 120]))} if g.random() < 0.4 else None
            yaw = {int(g.integers(1, K)): (int(g.integers(0, B)), float(g.choice([0.03, 0.06, 0.08])))} if g.random() < 0.4 else None
            seen, w = T._run_modes(B, K, wh, bwh, fx, seed, modes, rekey_at=rekey, short_list_seq=pick(), empty_kf_seq=pick(),
                                   use_lists=kw["use_lists"], voc_kL=(int(g.choice([4, 5, 8])), 5), few_points=few, yaw_error=yaw,
                                   defer_drop=bool(g.random() < 0.3), min_inliers=int(g.choice([0, 0, 50, 300, 100000])))
            for k_, v_ in seen.items():
                branches[k_] = branches.get(k_, 0) + v_
        else:
            w, stats = T._run(B, K, wh, bwh, fx, seed=seed, check_workload=False, **kw)
        worst = max(worst, w)
        frames += B * (K - 1)
    except AssertionError as e:
        bad.append(desc + " :: " + str(e)[:300])
    except NotImplementedError:
        skipped += 1
    runs += 1
    print("[%5.0f s] %d drives, %d tracked frames, %d mismatching drives, worst relative pose difference %.3g" % (time.time() - t0, runs, frames, len(bad), worst), flush=True)
print("RESULT: %d drives, %d tracked frames compared with the oracle chain, %d mismatching drives, worst relative pose difference %.3g" % (runs, frames, len(bad), worst))
print("reference-key-frame branches seen (sequence-frames):", branches, "; drives the lockstep (granular) driver cannot represent:", skipped)
for b in bad[:10]:
    print("MISMATCH", b)
