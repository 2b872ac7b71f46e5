#!/usr/bin/env python3
"""Searches the slot table of the packed S_t^-1 (Layout::kSinvSlot in csrc/a1mpc_solver.hpp): a bijection of the 78 entries (i >= j) of the
symmetric 12x12 onto the 78 doubles of a horizon step's S area such that the twelve entries a backward-sweep read touches -- lane ci reads
entry (max(ci, b), min(ci, b)) for b = 0..11, i.e. "cross" b -- lie on twelve different residues mod 16.  A ds_read_b64 is served in two
32-lane groups on 64 dword banks, and the lane group of the twin rows holds TWO QPs whose images are 16 doubles (mod 32) apart: distinct
residues mod 16 inside a QP make the 24 reads of a group conflict-free (row-major triangular packing: 21 LDS cycles per 12 reads, PMC
SQ_LDS_BANK_CONFLICT 11.7 % of the ADMM kernel's LDS cycles in round 2).  Secondary (soft) objective: the factor pass's stores
(lane ci writes entry (ci, min(b, ci)), ds_write_b64: 16-lane groups, 16 double banks).  Simulated annealing; prints the C table.
usage: sinv_layout_search.py [seed]"""
import random, sys
pairs = [(i, j) for i in range(12) for j in range(i + 1)]
pid = {p: k for k, p in enumerate(pairs)}
cross = [[pid[(max(b, c), min(b, c))] for c in range(12)] for b in range(12)]
store = [[pid[(c, min(b, c))] for c in range(12)] for b in range(12)]
def clashes(perm, sets): return sum(12 - len(set(perm[k] % 16 for k in s)) for s in sets)
def cost(perm): return 100 * clashes(perm, cross) + clashes(perm, store)
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
perm = list(range(78)); random.shuffle(perm)
c, T = cost(perm), 1.0
for it in range(2000000):
    a, b = random.randrange(78), random.randrange(78)
    perm[a], perm[b] = perm[b], perm[a]
    c2 = cost(perm)
    if c2 <= c or random.random() < 2.718 ** ((c - c2) / T): c = c2
    else: perm[a], perm[b] = perm[b], perm[a]
    T = max(0.03, T * 0.99995)
    if c <= 8: break   # 8 store clashes are the floor the search has ever reached with clash-free reads
print("// read clashes %d, store clashes %d (tools/sinv_layout_search.py)" % (clashes(perm, cross), clashes(perm, store)))
for i in range(12):
    print("    " + " ".join("%2d," % perm[pid[(i, j)]] for j in range(i + 1)))
