"""Per-iteration durations of the wavefront stage kernels from a rocprofv3 kernel-trace database:
extend / shadow / path / bsdf in ms for every iteration of the uncounted kernels, `step` iterations per line."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
def pick(tag): return [(s, (e - s) / 1e6) for n, s, e in rows if tag in n]
ext, sh, pa, bs, pl = pick('wf_extend_kernel<false'), pick('wf_shadow_kernel<false>'), pick('wf_pathhit_kernel<false'), pick('wf_bsdf_kernel<false>'), pick('wf_plan_kernel')
print('launches', len(ext), 'extend total %.1f shadow %.1f path %.1f bsdf %.1f plan %.1f ms' % (sum(d for _, d in ext), sum(d for _, d in sh), sum(d for _, d in pa), sum(d for _, d in bs), sum(d for _, d in pl)))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = min(len(ext), len(sh), len(pa), len(bs))
for i in range(0, n, step):
    print(i, ' '.join('%.2f/%.2f/%.2f/%.2f' % (ext[j][1], sh[j][1], pa[j][1], bs[j][1]) for j in range(i, min(i + step, n))))
