"""Bank-conflict model of the scatter / gather tiles for the benchmark scene (CPU only, numpy).

Bins the S-grip particles exactly as smac_sort.hpp does (block -> rank-in-cell bins -> cell order inside a bin -> chunks of
256 -> waves of 64) and counts, with the rules of MI355X_MICROARCH.md (LDS):
   ds_add_u32  : two groups of 32 lanes, bank = word index mod 32, a group costs max multiplicity of DISTINCT addresses on a bank
   ds_read_b128: four 16-lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32), bank row of 64 dwords -> a 16-B record at index w
                 occupies banks 4 (w mod 16) ..; a group costs the max number of distinct records on one 16-B slot
   ds_read_b96 : eight 8-lane groups, bank = (a/4) mod 32 -> 16-B record w sits on slot w mod 8
for tile strides (TSX, TSY).   python tools/lds_bank_model.py [n_particles]
"""
import sys
import numpy as np
sys.path.insert(0, ".")
from softmac_amd import scenes

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
n = 128 if N >= 1 << 20 else 64
cfg, env_dt, state, specs, s13 = scenes.s_grip(N, n, 8, "float32", 0, seed=1)
x = state[:, :3]
base = np.floor(x * n - 0.5).astype(np.int64)
base = np.clip(base, 0, n - 3)
nb = n // 4
blk = ((base[:, 0] >> 2) * nb + (base[:, 1] >> 2)) * nb + (base[:, 2] >> 2)
loc = ((base[:, 0] & 3) << 4) | ((base[:, 1] & 3) << 2) | (base[:, 2] & 3)
cell = blk * 64 + loc
# rank in cell by arrival order (the device uses atomics: any order, same statistics)
order = np.argsort(cell, kind="stable")
cs = cell[order]
first = np.r_[True, cs[1:] != cs[:-1]]
idx = np.arange(len(cs))
start = np.maximum.accumulate(np.where(first, idx, 0))
rank = np.empty(len(cs), dtype=np.int64)
rank[order] = idx - start
KMAX = 16
key = blk * KMAX + np.minimum(rank, KMAX - 1)
perm = np.lexsort((loc, key))            # inside a bin: by cell
blk_s, loc_s = blk[perm], loc[perm]
fx = (x * n - base)[perm]                # (unused: the stencil base IS the cell)
# chunks: every block cut into pieces of 256
bfirst = np.r_[True, blk_s[1:] != blk_s[:-1]]
bstart = np.maximum.accumulate(np.where(bfirst, np.arange(len(blk_s)), 0))
within = np.arange(len(blk_s)) - bstart
chunk_id = np.cumsum(bfirst | (within % 256 == 0)) - 1
cfirst = np.r_[True, chunk_id[1:] != chunk_id[:-1]]
cstart = np.maximum.accumulate(np.where(cfirst, np.arange(len(blk_s)), 0))
lane_in_chunk = np.arange(len(blk_s)) - cstart
wave = chunk_id * 4 + lane_in_chunk // 64
lane = lane_in_chunk % 64
nw = wave.max() + 1
print(f"N {N}  chunks {chunk_id.max() + 1}  waves {nw}  mean lanes/wave {len(wave) / nw:.1f}")
lx, ly, lz = (loc_s >> 4) & 3, (loc_s >> 2) & 3, loc_s & 3      # the stencil's first node is the cell itself: tile coords 0..3 (+0..2)

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 = G128 + [[l + 32 for l in g] for g in G128]
G96 = [[0, 1, 2, 3, 20, 21, 22, 23], [4, 5, 6, 7, 16, 17, 18, 19], [8, 9, 10, 11, 28, 29, 30, 31], [12, 13, 14, 15, 24, 25, 26, 27]]
G96 = G96 + [[l + 32 for l in g] for g in G96]
G32 = [list(range(32)), list(range(32, 64))]

def cost(word, groups, nbanks):
    """word: (nw, 64) tile index per lane (-1 = idle).  Sum over waves and groups of the max multiplicity of distinct words per bank."""
    total = 0
    for g in groups:
        w = word[:, g]                                  # (nw, L)
        L = w.shape[1]
        # distinct words only: sort along lanes, mark duplicates idle
        ws = np.sort(w, axis=1)
        dup = np.c_[np.zeros((len(ws), 1), bool), ws[:, 1:] == ws[:, :-1]]
        ws = np.where(dup, -1, ws)
        bank = np.where(ws >= 0, ws % nbanks, -1)
        m = np.zeros(len(ws), dtype=np.int64)
        for b in range(nbanks):
            m = np.maximum(m, (bank == b).sum(1))
        total += m.sum()                                # cycles (a group with any active lane costs >= 1)
    return total

def table(tsx, tsy):
    word = np.full((nw, 64), -1, dtype=np.int64)
    word[wave, lane] = lx * tsx + ly * tsy + lz
    out = {}
    for name, groups, nbanks in (("ds_add_u32", G32, 32), ("ds_read_b128", G128, 16), ("ds_read_b96", G96, 8)):
        tot = 0
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    off = i * tsx + j * tsy + k
                    tot += cost(np.where(word >= 0, word + off, -1), groups, nbanks)
        ideal = 27 * nw * len(groups)
        out[name] = tot / ideal
    return out

for tsx, tsy in ((36, 6), (68, 8), (100, 8), (72, 12), (80, 12), (68, 10), (132, 8), (52, 8), (84, 8), (40, 6)):
    t = table(tsx, tsy)
    print(f"TSX {tsx:4d} TSY {tsy:3d}  words {6 * tsx:5d}   cycles / conflict-free:  " + "  ".join(f"{k} {v:.2f}" for k, v in t.items()))
