"""Monte-Carlo model of how the traversal kernel's waves use their lanes (no GPU needed).

A ray is a sequence of node steps and leaf visits (per node step it moves on to a leaf with probability
LEAF_PER_RAY / NODE_PER_RAY or ends; after a leaf it ends or descends again). A wave of 64 lanes carries R rays per
lane; every pass the lanes whose ray in hand is at a node take a node step, and a leaf pass is added when V lanes have
only leaf work left (or nobody can descend); lanes with a free place take new rays when `refill` of them are free.
With R = 1, V = 24, refill = 8 this reproduces the pass statistics measured on the kernel (tools/stamps.py): 43.8
descending lanes per pass, a leaf pass every 5.7 passes with 25.9 lanes. With R = 2 (a lane works on whichever of its two
rays fits the pass) it predicts 55 descending lanes and about 10 % less work per ray at V = 8, given the stated costs of
a node step, a leaf pass, an exchange of the two rays and a refill in vector instructions.

usage: tools/lane_model.py [passes]
"""
import random
import sys

NODE_PER_RAY, LEAF_PER_RAY = 16.2, 1.7          # measured on the bench scene (production walk)
P_LEAF = LEAF_PER_RAY / NODE_PER_RAY
P_END_NODE = 0.5 / NODE_PER_RAY                 # a ray ends after a node step (stack empty) or after a leaf, half and half
P_END_LEAF = 0.5 / LEAF_PER_RAY
NODE, LEAF, NONE = 0, 1, 2


def step_node(rng):
    r = rng.random()
    return LEAF if r < P_LEAF else (NONE if r < P_LEAF + P_END_NODE else NODE)


def step_leaf(rng):
    return NONE if rng.random() < P_END_LEAF else NODE


def simulate(R, V, refill=8, passes=20000, c_node=150, c_leaf=110, c_swap=20, c_refill=30, seed=1):
    rng = random.Random(seed)
    st = [[NONE] * R for _ in range(64)]
    act = [0] * 64
    node_work = leaf_work = leaf_passes = cost = rays = swaps = 0
    for _ in range(passes):
        blocked = sum(1 for s in st if LEAF in s and NODE not in s)
        can_descend = sum(1 for s in st if NODE in s)
        busy = sum(1 for s in st if any(x != NONE for x in s))
        need = min(V, max(1, busy // 2))
        leaf_on = blocked >= need or (can_descend == 0 and blocked > 0)
        pref = LEAF if leaf_on else NODE
        swapped = False
        for l in range(64):
            if st[l][act[l]] != pref and pref in st[l]:
                act[l] = st[l].index(pref)
                swapped = True
        if R > 1 and swapped:
            cost += c_swap
            swaps += 1
        free = sum(1 for s in st if NONE in s)
        do_refill = free >= refill
        filled = [False] * 64
        if do_refill:
            for l in range(64):
                if st[l][act[l]] != NONE and NONE in st[l]:
                    st[l][st[l].index(NONE)] = NODE        # the new ray goes to the place set aside
                    filled[l] = True
        nn = nl = 0
        for l in range(64):
            s = st[l][act[l]]
            if s == NODE:
                st[l][act[l]] = step_node(rng)
                nn += 1
                rays += st[l][act[l]] == NONE
            elif s == LEAF and leaf_on:
                st[l][act[l]] = step_leaf(rng)
                nl += 1
                rays += st[l][act[l]] == NONE
            elif s == NONE and do_refill and not filled[l]:
                st[l][act[l]] = NODE
        node_work += nn
        leaf_work += nl
        cost += c_node if nn else 0
        if leaf_on and nl:
            cost += c_leaf
            leaf_passes += 1
        if do_refill:
            cost += c_refill
    return {'rays_per_lane': R, 'leaf_vote': V, 'descending_lanes_per_pass': round(node_work / passes, 1),
            'passes_per_leaf_pass': round(passes / max(1, leaf_passes), 2), 'lanes_in_a_leaf_pass': round(leaf_work / max(1, leaf_passes), 1),
            'instructions_per_ray': round(cost / max(1, rays), 1), 'passes_with_an_exchange': round(swaps / passes, 2)}


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    print(simulate(1, 24, passes=n))
    for R in (2, 3):
        for V in (8, 16, 24):
            print(simulate(R, V, passes=n))
