"""TEST INFRASTRUCTURE ONLY - scalar restatement of the reference's 3-D stability decision for the QM9 atom set:
``get_bond_order`` (evaluation/bond_analyze.py:108-133, margins :85, tables :5-45) and the valence check of
``check_stability`` (evaluation/stability.py:40-73).  Pure-Python loops, one atom pair at a time, as the reference."""
import math

DECODER = ["H", "C", "N", "O", "F"]                                         # datasets/datasets_config.py:4
B1 = {"H": {"H": 74, "C": 109, "N": 101, "O": 96, "F": 92}, "C": {"H": 109, "C": 154, "N": 147, "O": 143, "F": 135},
      "N": {"H": 101, "C": 147, "N": 145, "O": 140, "F": 136}, "O": {"H": 96, "C": 143, "N": 140, "O": 148, "F": 142},
      "F": {"H": 92, "C": 135, "N": 136, "O": 142, "F": 142}}
B2 = {"C": {"C": 134, "N": 129, "O": 120}, "N": {"C": 129, "N": 125, "O": 121}, "O": {"C": 120, "N": 121, "O": 121}}
B3 = {"C": {"C": 120, "N": 116, "O": 113}, "N": {"C": 116, "N": 110}, "O": {"C": 113}}
M1, M2, M3 = 10, 5, 3
ALLOWED = {"H": 1, "C": 4, "N": 3, "O": 2, "F": 1}


def get_bond_order(a1, a2, distance):
    d = 100 * distance
    if d < B1[a1][a2] + M1:
        if a1 in B2 and a2 in B2[a1] and d < B2[a1][a2] + M2:
            if a1 in B3 and a2 in B3[a1] and d < B3[a1][a2] + M3:
                return 3
            return 2
        return 1
    return 0


def check_stability(positions, atom_type):
    """positions: list of (x, y, z); atom_type: list of ints -> (molecule_stable, nr_stable_atoms, n_atoms, orders)."""
    n = len(atom_type)
    nr = [0] * n
    orders = [[0] * n for _ in range(n)]
    for i in range(n):
        for j in range(i + 1, n):
            dist = math.sqrt(sum((positions[i][k] - positions[j][k]) ** 2 for k in range(3)))
            o = get_bond_order(DECODER[atom_type[i]], DECODER[atom_type[j]], dist)
            nr[i] += o
            nr[j] += o
            orders[i][j] = orders[j][i] = o
    stable = sum(int(ALLOWED[DECODER[t]] == b) for t, b in zip(atom_type, nr))
    return stable == n, stable, n, orders
