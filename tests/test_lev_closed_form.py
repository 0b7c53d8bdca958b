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
