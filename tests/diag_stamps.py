"""Diagnostic (not a test): kernel timeline of one warm-start iteration at config 2.
Build the library with `make -C bayesfmmm_amd/csrc clean && make -C bayesfmmm_amd/csrc EXTRA=-DBFMMM_TIMELINE` first."""
import sys; sys.path.insert(0, '.')
import numpy as np, bayesfmmm_amd as bf
from bench import make_config2
w = make_config2()
cfg = bf.default_config(model=0, K=3, n_eigen=6, basis_degree=3, tot_mcmc_iters=100)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
smp.set_state(**w["state"])
smp.run(bf.SWEEP_WARM, 50)
st = np.array(smp.get_state("stamps")) * 0.01
names = ["curve_z", "pair_gram", "pg_reduce", "factor", "sweep", "curve_chi", "loglik"]
t0 = st[0]
prev_end = None
for k, nm in enumerate(names):
    b, e = st[2 * k] - t0, st[2 * k + 1] - t0
    gap = "" if prev_end is None else " (gap after previous end %.2f us)" % (b - prev_end)
    print("%-10s start %8.2f end %8.2f  busy %6.2f us, last WG starts at +%.2f%s" % (nm, b, e, e - b, st[16 + k] - st[2 * k], gap))
    prev_end = e




print("pair_gram staging: loads issued %.2f, W stored (first data back) %.2f" % (st[47]-st[2], st[48]-st[2]))
print("pair_gram WG(0,0), rel. kernel start: begin %.2f  staged %.2f  sync %.2f  pairs %.2f  mfma+store %.2f | pi_alpha job %.2f" % tuple(st[[40,41,42,43,44,46]]-st[2]))
st_ = np.array(smp.get_state("stamps"))
print("sweep: setup %.2f us, loop %.2f us, tail %.2f us" % ((st_[24]-st_[8])*0.01, (st_[25]-st_[24])*0.01, (st_[9]-st_[25])*0.01))
print("sweep clocks/step  A-wave0: between %d  P1 %d  barrier %d  P2 %d  barrier %d" % tuple(st_[26:31]/21))
print("sweep clocks/step  last B-wave: between %d  pre-poll %d  poll %d  rest-of-step %d  barrier %d" % tuple(st_[48:53]/21))
print("sweep clocks/step  B-wave 4  : between %d  pre-poll %d  poll %d  rest-of-step %d  barrier %d" % tuple(st_[54:59]/21))
try:
    tr = np.array(smp.get_state("wgtrace")).reshape(-1, 3)
    nwg = 250
    t0 = tr[:nwg, 0].min()
    order = np.argsort(tr[:nwg, 0])
    print("pair_gram per-WG trace (latest 12 starters): wg (ct,ks) start end xcc se/cu-bits")
    for w in order[-12:]:
        meta = int(tr[w, 1]); xcc = meta >> 32; hw = meta & 0xffffffff
        print("  wg %3d (ct %d, ks %2d) start +%6.2f end +%6.2f xcc %d hw_id 0x%08x cu %d sh %d se %d" % (w, w % 10, w // 10, (tr[w, 0] - t0) * 0.01, (tr[w, 2] - t0) * 0.01, xcc, hw, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7))
    for x in range(8):
        sel = [w for w in range(nwg) if (int(tr[w, 1]) >> 32) == x and w % 10 != 9]
        if sel:
            print("  xcc %d: %2d real WGs, start min +%.2f max +%.2f, end min +%.2f max +%.2f, dur min %.2f max %.2f" % (x, len(sel), (tr[sel, 0].min() - t0) * 0.01, (tr[sel, 0].max() - t0) * 0.01,
                  (tr[sel, 2].min() - t0) * 0.01, (tr[sel, 2].max() - t0) * 0.01, ((tr[sel, 2] - tr[sel, 0]).min()) * 0.01, ((tr[sel, 2] - tr[sel, 0]).max()) * 0.01))
    import collections
    cnt = collections.Counter((int(tr[w, 1]) >> 32, (int(tr[w, 1]) >> 8) & 0xff) for w in range(nwg) if w % 10 != 9)
    print("max real WGs sharing one (xcc, se/sh/cu):", max(cnt.values()), " distinct CUs used:", len(cnt))
except Exception as ex:
    print("no wgtrace:", ex)
print("abs: stamps pair_gram start %.2f  latest-start %.2f  end %.2f | trace min start %.2f max start(all) %.2f max end(all) %.2f" % (st[2], st[17], st[3], t0 * 0.01, tr[:nwg, 0].max() * 0.01, tr[:nwg, 2].max() * 0.01))
w = int(np.argmax(tr[:nwg, 0])); print("latest starter overall: wg", w, "ct", w % 10, "ks", w // 10, "xcc", int(tr[w, 1]) >> 32)
w = int(np.argmax(tr[:nwg, 2])); print("latest finisher overall: wg", w, "ct", w % 10, "ks", w // 10, "xcc", int(tr[w, 1]) >> 32, "start +%.2f" % ((tr[w, 0] - t0) * 0.01))
tr_all = np.array(smp.get_state("wgtrace")).reshape(-1, 3)
nz = np.nonzero(tr_all[:, 0])[0]
print("trace entries written:", len(nz), "max wg id", nz.max())
late = [w for w in nz if tr_all[w, 0] > t0 + 500]
print("entries starting > 5 us after t0:", late[:20], [(round((tr_all[w,0]-t0)*0.01,2), round((tr_all[w,2]-t0)*0.01,2)) for w in late[:20]])
print("curve_z block 7 thread 0 clocks: load+stage %d | barrier %d | u_k %d | matvec %d | dots %d | unpack+gamma/lgamma %d | lgammas+log %d | quad+accept+store %d" % tuple(np.array(smp.get_state("stamps"))[32:40]))
try:
    zt = np.array(smp.get_state("ztrace")).reshape(-1, 3)[:512]
    z0 = zt[:, 0].min()
    dur = (zt[:, 2] - zt[:, 0]) * 0.01
    print("curve_z per-WG: start spread %.2f us, duration min %.2f median %.2f max %.2f us, last end +%.2f us" % ((zt[:, 0].max() - z0) * 0.01, dur.min(), np.median(dur), dur.max(), (zt[:, 2].max() - z0) * 0.01))
    for x in range(8):
        sel = [w for w in range(512) if (int(zt[w, 1]) >> 32) == x]
        if sel: print("  xcc %d: %3d WGs, dur median %.2f max %.2f, end max +%.2f" % (x, len(sel), np.median(dur[sel]), dur[sel].max(), (zt[sel, 2].max() - z0) * 0.01))
except Exception as ex:
    print("no ztrace:", ex)
try:
    zt = np.array(smp.get_state("ztrace")).reshape(-1, 3)[:512]
    tr = np.array(smp.get_state("wgtrace")).reshape(-1, 3)[:250]
    print("xcc of curve_z blocks 0..9:", [int(zt[w, 1]) >> 32 for w in range(10)], " pair_gram wgs 0..9:", [int(tr[w, 1]) >> 32 for w in range(10)])
    smp.run(bf.SWEEP_WARM, 1, first_iter=50)
    zt2 = np.array(smp.get_state("ztrace")).reshape(-1, 3)[:512]
    print("one more iteration -> curve_z blocks 0..9 on xcc:", [int(zt2[w, 1]) >> 32 for w in range(10)])
except Exception as ex:
    print("xcc probe failed:", ex)
try:
    zp = np.array(smp.get_state("zphase")).reshape(512, 8)
    names = ["load+stage", "barrier", "u_k", "matvec", "dots", "gamma+lgamma", "lgammas+log", "quad+accept+store"]
    tot = zp.sum(axis=1)
    order = np.argsort(tot)
    print("curve_z phase clocks over all 512 WGs (thread 0): total min %d median %d max %d" % (tot.min(), np.median(tot), tot.max()))
    for q, nm in enumerate(names):
        print("  %-18s min %6d  median %6d  max %6d   | in the 10 slowest WGs: median %6d" % (nm, zp[:, q].min(), np.median(zp[:, q]), zp[:, q].max(), np.median(zp[order[-10:], q])))
except Exception as ex:
    print("no zphase:", ex)
print("factor WG1 (us from its start): loads issued+zv %.2f | barrier %.2f | band products done %.2f | r reduce %.2f | Prec built %.2f | factor_core %.2f" % tuple((st[49:55]-st[48])))
fc = np.array(smp.get_state("fct")) * 0.01
print("factor_core: cholesky %.2f us | back-substitution %.2f | barrier %.2f | MFMA C + stores %.2f (then L copy)" % (fc[0]-st[53], fc[1]-fc[0], fc[2]-fc[1], fc[3]-fc[2]))
print("curve_chi: hyper job ends at +%.2f us of the kernel (kernel busy %.2f)" % (st[58]-st[10], st[11]-st[10]))
