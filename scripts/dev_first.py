import sys, time, numpy as np
sys.path.insert(0, '.')
from oracle import oracle_py as O
import snesimage_amd as S
img = O.synth_image(0x5EED0000)
oi = O.OracleImage(img, 8, 15)
oi.initialize_tiles(); oi.recalculate_palettes()
gi = S.OptimizedImage(img, 8, 15)
gi.tile_palettes = oi.tile_palettes
gi.palette = oi.palette
gi.optimize()
print("map equal:", np.array_equal(gi.palette_map, oi.palette_map))
e_o = oi.error(); e_g = gi.error()
print("error oracle", repr(e_o), "gpu", repr(e_g), "rel", abs(e_o-e_g)/e_o)
cands = O.random_candidates(1, 0, 16)
eo = oi.score_candidates(2, 3, cands); eg = gi.score_candidates(2, 3, cands)
print("cand rel max", np.max(np.abs(eo-eg)/eo), eo[:3], eg[:3])
# kmeans on gpu
g2 = S.OptimizedImage(img, 8, 15)
g2.initialize_tiles()
o2 = O.OracleImage(img, 8, 15); o2.initialize_tiles()
print("init tiles equal:", np.array_equal(g2.tile_palettes, o2.tile_palettes), np.array_equal(g2.palette, o2.palette), np.array_equal(g2.palette_map, o2.palette_map))
g2.recalculate_palettes(); o2.recalculate_palettes()
print("recalc equal:", np.array_equal(g2.palette, o2.palette), np.array_equal(g2.palette_map, o2.palette_map))
print("json equal:", g2.as_json() == o2.as_json())
# step
r_o = o2.step(0, 1, 2, 0, 1, 7); r_g = g2.step(0, 1, 2, 0, 1, 7)
print("step", r_o, r_g)
# timing
gi.set_chunk(256)
import ctypes
for n in (64, 256, 1024):
    c = O.random_candidates(2, n, n)
    gi.score_candidates(0, 0, c)
    t = time.time(); gi.score_candidates(0, 0, c); dt = time.time() - t
    print(n, "cands", dt*1e3, "ms", n/dt, "cand/s")
