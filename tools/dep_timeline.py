"""Timeline of the dependent schedule from a rocprofv3 kernel trace: per step, when the Phi kernel, the reduce and the ELBO launch start and end
relative to the previous ELBO launch's end.  usage: python tools/dep_timeline.py <kernel_trace.csv> [n_last=40]"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
def pick(sub):
    return [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if sub in r["Kernel_Name"]]
elbo = pick("elbo_chains_mfma_kernel")
phi = pick("phi_sort_kernel")
red = pick("phi_reduce_kernel")
print("launches: elbo %d phi %d reduce %d" % (len(elbo), len(phi), len(red)))
# the dependent schedule is the LAST block of launches in a --in-flight 0 run
E = elbo[-n_last:]
d_elbo = np.array([b - a for a, b in E]) / 1e3
gap = np.array([E[i + 1][0] - E[i][1] for i in range(len(E) - 1)]) / 1e3
period = np.array([E[i + 1][0] - E[i][0] for i in range(len(E) - 1)]) / 1e3
print("ELBO kernel duration us: median %.1f min %.1f max %.1f" % (np.median(d_elbo), d_elbo.min(), d_elbo.max()))
print("end of ELBO i -> start of ELBO i+1 us: median %.1f min %.1f max %.1f" % (np.median(gap), gap.min(), gap.max()))
print("period us: median %.1f" % np.median(period))
P = [p for p in phi if p[0] >= E[0][0] - 200000]
d_phi = np.array([b - a for a, b in P]) / 1e3
print("Phi kernel duration us (same window): median %.1f min %.1f max %.1f  n=%d" % (np.median(d_phi), d_phi.min(), d_phi.max(), len(P)))
# for each ELBO launch: offset of the Phi kernel that starts during / after it
offs = []
for a, b in E[:-1]:
    later = [p for p in P if p[0] >= a - 30000]
    if later:
        offs.append(((later[0][0] - a) / 1e3, (later[0][1] - a) / 1e3))
if offs:
    o = np.array(offs)
    print("Phi kernel of step i+1 relative to ELBO start of step i: starts %+.1f us, ends %+.1f us (medians)" % (np.median(o[:, 0]), np.median(o[:, 1])))
