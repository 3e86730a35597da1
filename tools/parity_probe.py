"""Which of the 17 buffers differ from the oracle, and where, at four small sizes (vector and scalar paths): the first thing to run on a
GPU box after a kernel change.  python tools/parity_probe.py"""
import numpy as np, sys
sys.path.insert(0, '.')
from miniengineao_amd import synth
from tests import helpers as H
from oracle import oracle as O
for (w, h) in [(256, 128), (322, 182), (512, 300), (136, 72)]:
    s = H.settings(O, w, h)
    depth = synth.make("S2", w, h, seed=5)
    want = O.run(depth, s)
    ao = H.component(s)
    got = ao.render(depth)
    bad = []
    if not np.array_equal(got, want["result"]): bad.append(("result", int((got != want["result"]).sum()), tuple(np.argwhere(got != want["result"])[0])))
    for i in H.valid_debug_ids(4):
        g = ao.debug_buffer(i); wv = want[H.NAMES[i]]
        if not np.array_equal(g, wv):
            idx = np.argwhere(g != wv)
            bad.append((H.NAMES[i], len(idx), tuple(idx[0]), tuple(idx[-1])))
    print((w, h), "BAD:", bad)
    ao.close()
