"""Per-wavefront timeline of one base shortlist launch (developer library, CHB_SL_DBG=<file>): how long do the
workgroups run, how many tiles do they compute, how busy are the CUs?   usage: python tools/sl_timeline.py file"""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 4)       # [workgroup][wavefront][start, end, tiles, hw id]
live = d[:, 0, 1] > 0
d = d[live]
t0 = d[:, :, 0].min()
st = (d[:, :, 0].min(axis=1) - t0) / 100.0          # us (100 MHz)
en = (d[:, :, 1].max(axis=1) - t0) / 100.0
dur = en - st
tiles = d[:, :, 2].astype(np.int64)                 # per wavefront
hw = d[:, 0, 3].astype(np.int64)
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; xcc = (hw >> 28) & 0xF   # (gfx9 HW_ID layout; XCC_ID in its own reg on gfx94x)
print(f"{len(d)} workgroups, launch span {en.max():.1f} us, first start {st.min():.1f}, last start {st.max():.1f}")
print("workgroup duration us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f  mean %.1f" %
      (dur.min(), *np.percentile(dur, [10, 50, 90]), dur.max(), dur.mean()))
tw = tiles.max(axis=1)
print("tiles computed (busiest wavefront of a workgroup): min %d median %d max %d;  mean over wavefronts %.1f" %
      (tw.min(), np.median(tw), tw.max(), tiles.mean()))
print("us per computed tile (duration / busiest wavefront's tiles): median %.3f" % np.median(dur / np.maximum(tw, 1)))
c = np.corrcoef(dur, tw)[0, 1]
print(f"correlation(duration, tiles) = {c:.3f}")
# duration against tiles, by decile of tiles
order = np.argsort(tw)
for k in range(10):
    sel = order[k * len(order) // 10:(k + 1) * len(order) // 10]
    print(f"  decile {k}: tiles {tw[sel].mean():7.1f}  duration {dur[sel].mean():7.1f} us")
# occupancy over time: how many workgroups are running
ts = np.linspace(0, en.max(), 41)
run = [(int(((st <= t) & (en > t)).sum())) for t in ts]
print("workgroups running over the launch span:", run)
key = hw & 0xFFFFFF00
print("distinct hardware ids (>> 8):", len(np.unique(key)))
