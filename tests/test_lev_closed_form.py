"""CPU: the closed form k_seed_search uses for bucket entries == the reference's table-driven LEV(1) automaton.

traverse_bursttrie.cpp:100-298 feeds the characteristic bit-vectors of the window's 9-mer P (partialwin chars) and the 10 chars
T of (trie path + bucket tail) through the universal Levenshtein-1 tables (:68-98) and accepts an entry at the first depth
>= partialwin-2 with state >= 8; state 9 at depth partialwin-1 is the 0-error match.  With a = longest common prefix of (P, T) and
s0 / s1 / s2 = the number of trailing equal characters of P vs T, P vs T shifted left by one, P shifted left by one vs T:
    accepted at depth pw-2  <=>  a + s2 >= pw-1        (edit distance(P, T[0..pw-1)) <= 1 : a deletion)
    else at depth pw-1      <=>  a + s0 >= pw-1        (at most one substitution in the first pw chars)
    else at depth pw        <=>  a + s1 >= pw          (one insertion)
    0-error match           <=>  a >= pw               (and then the entry was already accepted at depth pw-2)
The oracle's orc_lev_accepts() runs the tables; this test compares the closed form with it for every supported seed length on
structured (0, 1, 2 edits) and random pairs."""
import ctypes as C

import numpy as np
import pytest

from helpers import orc


def closed_form(P, T, pw):
    """-> (accepted, first accepting depth, zero) with the bit tricks of smr_seed.hpp::lev1_entry, on Python ints"""
    eq0 = [((P >> (2 * i)) & 3) == ((T >> (2 * i)) & 3) for i in range(pw)]
    eq1 = [((P >> (2 * i)) & 3) == ((T >> (2 * (i + 1))) & 3) for i in range(pw)]
    eq2 = [((P >> (2 * (i + 1))) & 3) == ((T >> (2 * i)) & 3) for i in range(pw - 1)]

    def lead(v):
        n = 0
        while n < len(v) and v[n]:
            n += 1
        return n

    a, s0, s1, s2 = lead(eq0), lead(eq0[::-1]), lead(eq1[::-1]), lead(eq2[::-1])
    c8, c9, c10 = a + s2 >= pw - 1, a + s0 >= pw - 1, a + s1 >= pw
    depth = pw - 2 if c8 else (pw - 1 if c9 else pw)
    return (c8 or c9 or c10), depth, a >= pw


def alive_closed_form(P, T, m, pw):
    """smr_seed.hpp::lev1_alive on Python ints: is the automaton alive after the first m chars of T?"""
    p = [(P >> (2 * i)) & 3 for i in range(pw)]
    t = [(T >> (2 * i)) & 3 for i in range(pw + 1)]

    def pe(i):
        return p[i] if 0 <= i < pw else -1

    a = 0
    while a < m and a < pw and p[a] == t[a]:
        a += 1
    if a >= m:
        return True
    return (all(pe(i) == t[i] for i in range(a + 1, m)) or all(pe(i - 1) == t[i] for i in range(a + 1, m))
            or all(pe(i + 1) == t[i] for i in range(a, m)))


@pytest.mark.parametrize("pw", [4, 5, 6, 7, 8, 9, 10])
def test_prefix_viability_equals_table_automaton(pw):
    """trie nodes: `state != 14` after the first m chars  <=>  lev1_alive(P, T, m), for every prefix length"""
    L = orc.lib()
    L.orc_lev_alive_depth.restype = C.c_uint32
    L.orc_lev_alive_depth.argtypes = [C.c_uint32] * 3
    rng = np.random.default_rng(200 + pw)
    for _ in range(6000):
        P = int(rng.integers(0, 4 ** pw))
        pl = [(P >> (2 * i)) & 3 for i in range(pw)]
        for kind in range(7):
            t = pl[:]
            if kind == 1:
                j = int(rng.integers(0, pw)); t[j] = (t[j] + int(rng.integers(1, 4))) & 3
            elif kind == 2:
                t.insert(int(rng.integers(0, pw + 1)), int(rng.integers(0, 4)))
            elif kind == 3:
                del t[int(rng.integers(0, pw))]
            elif kind == 4:
                j = int(rng.integers(0, pw)); t[j] = (t[j] + 1) & 3
                j = int(rng.integers(0, pw)); t[j] = (t[j] + 2) & 3
            elif kind == 5:
                del t[int(rng.integers(0, pw))]
                t.insert(int(rng.integers(0, pw)), int(rng.integers(0, 4)))
            elif kind == 6:
                t = [int(x) for x in rng.integers(0, 4, size=pw + 1)]
            while len(t) < pw + 1:
                t.append(int(rng.integers(0, 4)))
            T = sum(c << (2 * i) for i, c in enumerate(t[:pw + 1]))
            ad = L.orc_lev_alive_depth(P, T, pw)
            for m in range(1, pw + 2):
                assert (ad >= m) == alive_closed_form(P, T, m, pw), (pw, P, T, m, ad)


@pytest.mark.parametrize("pw", [4, 5, 6, 7, 8, 9, 10])
def test_closed_form_equals_table_automaton(pw):
    L = orc.lib()
    L.orc_lev_accepts.restype = C.c_uint32
    L.orc_lev_accepts.argtypes = [C.c_uint32] * 3
    rng = np.random.default_rng(100 + pw)
    n_acc = 0

    def check(P, T):
        nonlocal n_acc
        r = L.orc_lev_accepts(P, T, pw)
        acc, depth, zero = closed_form(P, T, pw)
        assert bool(r & 1) == acc, (pw, P, T, hex(r))
        if acc:
            n_acc += 1
            assert (r >> 8) == depth, (pw, P, T, hex(r), depth)
            assert bool(r & 2) == zero, (pw, P, T, hex(r))
            if zero:
                assert depth == pw - 2          # a 0-error match is always a COND candidate, never UNCOND

    for _ in range(12000):
        P = int(rng.integers(0, 4 ** pw))
        pl = [(P >> (2 * i)) & 3 for i in range(pw)]
        for kind in range(6):
            t = pl[:]
            if kind == 1:
                j = int(rng.integers(0, pw)); t[j] = (t[j] + int(rng.integers(1, 4))) & 3
            elif kind == 2:
                t.insert(int(rng.integers(0, pw + 1)), int(rng.integers(0, 4)))
            elif kind == 3:
                del t[int(rng.integers(0, pw))]
            elif kind == 4:
                j = int(rng.integers(0, pw)); t[j] = (t[j] + 1) & 3
                j = int(rng.integers(0, pw)); t[j] = (t[j] + 2) & 3
            elif kind == 5:
                del t[int(rng.integers(0, pw))]
                t.insert(int(rng.integers(0, pw)), int(rng.integers(0, 4)))
            while len(t) < pw + 1:
                t.append(int(rng.integers(0, 4)))
            check(P, sum(c << (2 * i) for i, c in enumerate(t[:pw + 1])))
    for _ in range(30000):
        check(int(rng.integers(0, 4 ** pw)), int(rng.integers(0, 4 ** (pw + 1))))
    assert n_acc > 20000


def pg_key(T, frm, cnt):
    """smr_host.hpp::pg_key: chars frm..frm+cnt-1 of a packed string as a number, first char most significant"""
    k = 0
    for q in range(cnt):
        k = (k << 2) | ((T >> (2 * (frm + q))) & 3)
    return k


def pigeonhole_reaches(P, T, pw, cA, cB):
    """smr_seed_pg.hpp: is candidate string T inside one of the four directory ranges k_seed_pg reads for pattern P, with directories
    over the first cA <= pw/2 chars (array EA) and over chars h..h+cB-1, cB <= pw-h (array EB)?"""
    h = pw // 2
    if pg_key(T, 0, cA) == pg_key(P, 0, cA):                       # A
        return True
    kt = pg_key(T, h, cB)
    if kt == pg_key(P, h, cB) or kt == pg_key(P, h - 1, cB):       # S0, S1
        return True
    if cB == pw - h:                                               # S2: T[pw-1] is free
        return (kt >> 2) == pg_key(P, h + 1, cB - 1)
    return kt == pg_key(P, h + 1, cB)


@pytest.mark.parametrize("pw", [4, 5, 6, 7, 8, 9, 10])
def test_every_accepted_string_lies_under_one_of_the_four_exact_keys(pw):
    """the completeness of the pigeonhole search: whatever lev1_entry accepts is found through key A, S0, S1 or S2, for every
    directory width the index builder may choose (smr_host.hpp::pg_chars)"""
    rng = np.random.default_rng(300 + pw)
    h = pw // 2
    widths = [(cA, cB) for cA in range(1, h + 1) for cB in range(1, pw - h + 1)]
    n_acc = 0
    for _ in range(6000 if pw < 9 else 3000):
        P = int(rng.integers(0, 4 ** pw))
        pl = [(P >> (2 * i)) & 3 for i in range(pw)]
        t = pl[:]
        kind = int(rng.integers(0, 6))
        if kind == 1:
            j = int(rng.integers(0, pw)); t[j] = (t[j] + int(rng.integers(1, 4))) & 3
        elif kind == 2:
            t.insert(int(rng.integers(0, pw + 1)), int(rng.integers(0, 4)))
        elif kind == 3:
            del t[int(rng.integers(0, pw))]
        elif kind == 4:
            j = int(rng.integers(0, pw)); t[j] = (t[j] + 1) & 3
            j = int(rng.integers(0, pw)); t[j] = (t[j] + 2) & 3
        elif kind == 5:
            t = [int(x) for x in rng.integers(0, 4, size=pw + 1)]
        while len(t) < pw + 1:
            t.append(int(rng.integers(0, 4)))
        T = sum(c << (2 * i) for i, c in enumerate(t[:pw + 1]))
        acc, _, _ = closed_form(P, T, pw)
        if acc:
            n_acc += 1
            for cA, cB in widths:
                assert pigeonhole_reaches(P, T, pw, cA, cB), (pw, P, T, cA, cB)
    assert n_acc > 1500


def test_the_four_keys_are_exhaustive_on_a_small_alphabet_of_lengths():
    """pw = 4 and 5 exhaustively: every (P, T) pair"""
    for pw in (4, 5):
        h = pw // 2
        for P in range(4 ** pw):
            for T in range(4 ** (pw + 1)):
                if closed_form(P, T, pw)[0]:
                    assert pigeonhole_reaches(P, T, pw, h, pw - h) and pigeonhole_reaches(P, T, pw, 1, 1), (pw, P, T)


def _accepted_per_pattern(Ps, pw):
    """how many of the 4^(pw+1) strings T lev1_entry accepts for each pattern of Ps (the bit form of smr_seed.hpp, vectorised)"""
    u = np.uint64
    T = np.arange(4 ** (pw + 1), dtype=np.uint64)[None, :]
    P = np.asarray(Ps, dtype=np.uint64)[:, None]
    m2 = u((1 << (2 * pw)) - 1)
    m2b = m2 >> u(2)
    x0, x1, x2 = (P ^ T) & m2, (P ^ (T >> u(2))) & m2, ((P >> u(2)) ^ T) & m2b
    y = x0 | u(1 << (2 * pw))
    a2 = np.log2((y & (~y + u(1))).astype(np.float64)).astype(np.uint64) & ~u(1)          # 2 * common prefix length
    return ((((x0 >> a2) >> u(2)) == 0) | ((x1 >> a2) == 0) | ((x2 >> a2) == 0)).sum(axis=1)


@pytest.mark.parametrize("pw", [4, 5, 6, 7, 8, 9, 10])
def test_a_search_accepts_at_most_31_pw_minus_20_strings(pw):
    """What bounds a search's hit list (SEED_HCAP_BOUND in smr_engine_seed.hpp): every accepted string is one entry of the mini-trie = at most one id.
    Exhaustive over P for pw <= 6; for the longer seeds the patterns without equal neighbours (they reach the maximum) + random ones."""
    import os
    import re
    from helpers import paths
    bound = 31 * pw - 20
    if pw <= 6:
        c = np.concatenate([_accepted_per_pattern(np.arange(s, min(s + 256, 4 ** pw)), pw) for s in range(0, 4 ** pw, 256)])
        assert int(c.max()) == bound
    else:
        rng = np.random.Generator(np.random.PCG64(pw))
        pats = [sum(((i + s) & 3) << (2 * i) for i in range(pw)) for s in range(4)]               # ACGTACG... : no char equals a neighbour or a neighbour's neighbour
        pats += [sum(((i * k) & 3) << (2 * i) for i in range(pw)) for k in (1, 3)]
        pats += [int(x) for x in rng.integers(0, 4 ** pw, 6 if pw == 10 else 10)]
        c = np.concatenate([_accepted_per_pattern(pats[i:i + 2], pw) for i in range(0, len(pats), 2)])
        assert int(c.max()) == bound
    src = open(os.path.join(paths.REPO, "sortmerna_amd", "csrc", "smr_engine_seed.hpp")).read()
    assert re.search(r"#define SEED_HCAP_BOUND\(pw\) \(31u \* \(pw\) - 20u\)", src)
