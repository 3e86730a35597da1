import sys, time, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O
from miniengineao_amd import synth
from tests import helpers as H
for (w,h) in [(67,45),(256,256),(1920,1080)]:
    s = H.settings(O, w, h); d = synth.make('S2', w, h)
    want = O.run(d, s, nthreads=8)
    ao = H.component(s)
    got = ao.render(d)
    print(w,h,'result equal', np.array_equal(got, want['result']), 'ndiff', int((got!=want['result']).sum()))
    for i in H.valid_debug_ids(4):
        g = ao.debug_buffer(i); print('  ', i, H.NAMES[i], np.array_equal(g, want[H.NAMES[i]]), int((g!=want[H.NAMES[i]]).sum()))
    ao.close()
