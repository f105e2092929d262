#!/usr/bin/env python3
"""Accuracy + wall-clock of the whole CLI pipeline (frag -> train -> predict -> eval) on a synthetic community with a
phylogeny: G genera x S species x T strains; strains of a species differ by strain_div (default 1 %), species of a genus by species_div (default 8 %), genera are
unrelated.  Training sees T-1 strains of every species, testing draws fragments from the held-out strain (the Vervier
set-up: classify reads of unseen strains at species level).  GPU box only; scratch files in /tmp."""
import os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opal_amd import opal

G, S, T, glen = (int(x) for x in (sys.argv[1:5] + ["8", "8", "4", "200000"][len(sys.argv) - 1:])[:4])
num_hash = int(sys.argv[5]) if len(sys.argv) > 5 else 2
species_div = float(sys.argv[6]) if len(sys.argv) > 6 else 0.08
strain_div = float(sys.argv[7]) if len(sys.argv) > 7 else 0.01
rng = np.random.default_rng(7)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def mutate(g, rate):
    g = g.copy()
    m = rng.random(g.size) < rate
    g[m] = (g[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
    return g


d = tempfile.mkdtemp(prefix="opal_acc_", dir="/tmp")
os.makedirs(d + "/train"); os.makedirs(d + "/test")
ftr, ttr = open(d + "/train/A.fasta", "w"), open(d + "/train/A.taxid", "w")
fte, tte = open(d + "/test/A.fasta", "w"), open(d + "/test/A.taxid", "w")
for gi in range(G):
    genus = rng.integers(0, 4, size=glen, dtype=np.uint8)
    for si in range(S):
        species = mutate(genus, species_div)
        taxid = "%d" % (1000 * (gi + 1) + si)
        for ti in range(T):
            strain = bytes(ACGT[mutate(species, strain_div)])
            f, t = (fte, tte) if ti == T - 1 else (ftr, ttr)
            f.write(">g%d_s%d_t%d\n" % (gi, si, ti))
            f.write("\n".join(strain[j:j + 80].decode() for j in range(0, glen, 80)) + "\n")
            t.write(taxid + "\n")
for h in (ftr, ttr, fte, tte):
    h.close()
np.random.seed(3)
t0 = time.time()
opal.main(["simulate", d + "/test", d + "/train", d + "/out", "-r", "-l", "200", "-c", "2", "--num-hash", str(num_hash),
           "--num-batches", "3", "--bits", "30"])
print("community: %d genera x %d species x %d strains of %d bp, species divergence %.3f, strain divergence %.3f, numHash %d; K=%d; "
      "pipeline wall %.1f s" % (G, S, T, glen, species_div, strain_div, num_hash, G * S, time.time() - t0))
print(open(d + "/out/2model/vw-model_vwps.log").read())
shutil.rmtree(d)
