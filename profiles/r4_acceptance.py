import sys, numpy as np
sys.path.insert(0, '/root/repo')
import snesimage_amd as S
from snesimage_amd.synth import synth_image
img = synth_image(0x5EED0000)
im = S.OptimizedImage(img, 8, 15)
im.initialize_tiles(); im.recalculate_palettes()
slots = S.schedule(8, 15, 410)
acc = []
prev = im.palette.copy()
for i in range(410):
    m, p, idx, ch, _ = slots[i]
    im.step(S.METHOD_RANDOM, p, idx, ch, 1, i, 4096)
    cur = im.palette
    acc.append(int(not np.array_equal(cur, prev))); prev = cur.copy()
a = np.array(acc)
print('acceptance per 50 calls:', [float(a[i:i+50].mean()) for i in range(0, 400, 50)])
