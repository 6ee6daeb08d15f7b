"""Aggregate BA iterations/s when S independent windows are solved concurrently (one context = one stream + one graph per
window, one host thread each): how much of the GPU the latency-bound single-window solve leaves unused."""
import sys, threading, time
sys.path.insert(0, '.')
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate

def run(S, vio=True, steps=100):
    pbs = [synth.make_window(n_frames=10, n_landmarks=1000, use_inertial=vio, preintegrate=preintegrate if vio else None) for _ in range(1)]
    ctxs = [HipContext(device=0) for _ in range(S)]
    for c in ctxs:
        c.upload(pbs[0])
        for _ in range(5):
            c.solve_resident(BASummary(pbs[0], trace=False))
    iters = [0] * S
    def work(i):
        sm = BASummary(pbs[0], trace=False)
        for _ in range(steps):
            ctxs[i].solve_resident(sm)
            iters[i] += sm.num_iterations
    th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    for c in ctxs: c.close()
    return sum(iters) / dt

for vio in (True, False):
    print('vio' if vio else 'vision', {S: round(run(S, vio)) for S in (1, 2, 3, 4, 6, 8)})
