"""Per-iteration durations of the wavefront stage kernels from a rocprofv3 kernel-trace database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
def pick(tag): return [(s, (e - s) / 1e6) for n, s, e in rows if tag in n]
ext, sh, bs, pl = pick('wf_extend_kernel<false'), pick('wf_shade_kernel<false>'), pick('wf_bsdf_kernel<false>'), pick('wf_plan_extend')
print('launches', len(ext), 'extend total %.1f shade %.1f bsdf %.1f plan %.1f ms' % (sum(d for _, d in ext), sum(d for _, d in sh), sum(d for _, d in bs), sum(d for _, d in pl)))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for i in range(0, len(ext), step):
    print(i, ' '.join('%.2f/%.2f/%.2f' % (ext[j][1], sh[j][1], bs[j][1]) for j in range(i, min(i + step, len(ext)))))
