// TEST INFRASTRUCTURE: the emulator's stand-in for sortmerna_amd/csrc/smr_device_ops.hpp (found first: -I tests/emu/shim comes first).
// Scalar host code with the semantics of the gfx950 instructions the product header maps to.
#pragma once
#include <stdint.h>

#define SMR_DYN_LDS(type, name) type* const name = (type*)emu::dyn_lds()
#define SMR_GLOBAL_U32 const uint32_t
#define SMR_SW_SELFCHECK_CASES 8u

namespace smr {

struct pk16 { int16_t lo, hi; };
inline pk16 pk_from(uint32_t v) { pk16 r; r.lo = (int16_t)(v & 0xFFFF); r.hi = (int16_t)(v >> 16); return r; }
inline uint32_t pk_bits(pk16 v) { return (uint32_t)(uint16_t)v.lo | ((uint32_t)(uint16_t)v.hi << 16); }
inline pk16 pk_add(pk16 a, pk16 b) { pk16 r; r.lo = (int16_t)(a.lo + b.lo); r.hi = (int16_t)(a.hi + b.hi); return r; }      // v_pk_add_i16 (wraps)
inline pk16 pk_sub(pk16 a, pk16 b) { pk16 r; r.lo = (int16_t)(a.lo - b.lo); r.hi = (int16_t)(a.hi - b.hi); return r; }
inline pk16 pk_subs_u(pk16 a, pk16 b) { pk16 r; const uint16_t al = (uint16_t)a.lo, bl = (uint16_t)b.lo, ah = (uint16_t)a.hi, bh = (uint16_t)b.hi;      // v_pk_sub_u16 clamp
  r.lo = (int16_t)(al > bl ? al - bl : 0); r.hi = (int16_t)(ah > bh ? ah - bh : 0); return r; }
inline pk16 pk_max(pk16 a, pk16 b) { pk16 r; r.lo = a.lo > b.lo ? a.lo : b.lo; r.hi = a.hi > b.hi ? a.hi : b.hi; return r; }
inline uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel) {          // v_perm_b32: bytes 0-3 = s1, 4-7 = s0, 8-11 = sign of a 16-bit half, 12 = 0x00, >= 13 = 0xFF
  const unsigned long long src = ((unsigned long long)s0 << 32) | s1;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t s = (sel >> (8 * i)) & 0xFF;
    uint32_t b;
    if (s < 8) b = (uint32_t)(src >> (8 * s)) & 0xFF;
    else if (s < 12) b = ((src >> (8 * (2 * (s - 8) + 1) + 7)) & 1) ? 0xFF : 0x00;
    else if (s == 12) b = 0x00;
    else b = 0xFF;
    r |= b << (8 * i);
  }
  return r;
}
inline uint32_t div_multiple(uint32_t n, uint32_t d) { return n / d; }
inline uint32_t wave_scan_add(uint32_t x) {                 // the product's six v_add_u32_dpp, step by step through the emulator's DPP
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
  return x;
}
// uni(): v_readfirstlane_b32 of a value the kernel claims to be wave-uniform -- here the claim is CHECKED: every active lane must hold the
// value of the first one
inline uint32_t uni(uint32_t v) {
  const unsigned long long m = __ballot(1);
  const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_ctzll(m));
  if (!__all(f == v)) { fprintf(stderr, "emu: uni() of a value that differs between the lanes of a wave (%u vs %u)\n", f, v); abort(); }
  return f;
}
inline int uni(int v) { return (int)uni((uint32_t)v); }
inline unsigned long long uni(unsigned long long v) { return (unsigned long long)uni((uint32_t)v) | ((unsigned long long)uni((uint32_t)(v >> 32)) << 32); }
inline unsigned long uni(unsigned long v) { return (unsigned long)uni((unsigned long long)v); }
template <class T> inline T uni_words(const T& v) {
  static_assert(sizeof(T) % 4 == 0, "uni_words: whole words only");
  uint32_t w[sizeof(T) / 4];
  __builtin_memcpy(w, &v, sizeof(T));
  for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = uni(w[i]);
  T o;
  __builtin_memcpy(&o, w, sizeof(T));
  return o;
}

}  // namespace smr
