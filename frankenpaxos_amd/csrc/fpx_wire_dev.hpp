// fpx_wire_dev.hpp -- the wire adapter's decoders on the device (include/fpx_wire.h, "_dev" entry points): a tick of
// serialised ProxyLeaderInbound / AcceptorInbound messages, already in HBM (or page-locked host memory), becomes the SoA
// batch the Phase-2 kernels take without the host parsing a byte.  One thread per message; the parser is the one of
// fpx_wire_parse.hpp the host decoders run, compiled for both sides, so the fields agree by construction and
// tests/test_wire_dev.py checks them against the google.protobuf vectors all the same.
//
// The work is byte-serial per message (a tag, a varint, ...), about 25 bytes for a Phase2a with a small command: a
// wavefront's 64 messages span ~1.6 KB, so its byte loads hit a handful of cache lines that the first touch brings in;
// HBM sees each byte once.  What bounds the stage is the copy of the tick into HBM, not this kernel.
#pragma once
#include "fpx_kernels.hpp"
#include "fpx_wire_parse.hpp"

namespace fpx {

enum { ST_WIRE = 5 };  // status word: 0x7fffffff - key of the first bad message, 0 = none

struct WireOut {
  int32_t *kind, *slot, *round, *is_noop, *value_len, *a, *b, *value_id;
  int64_t* value_off;
  int32_t value_id_base;
};

// key: offsets are judged before messages, as the host decoder does -- a bad offset outranks a malformed message
__device__ __forceinline__ void wire_bad(const State& st, bool parse, int32_t i) {
  atomicMax(&st.status[ST_WIRE], 0x7fffffff - ((parse ? 0x40000000 : 0) + i));
}

template <int WHICH>  // 0 ProxyLeaderInbound, 1 AcceptorInbound
__global__ void __launch_bounds__(256) k_wire_decode(const State st, const uint8_t* __restrict__ buf, int64_t buf_len,
                                                     const int64_t* __restrict__ offsets, int32_t n, const WireOut o) {
  const int32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  bool bounds = fpxw::offset_ok(offsets, i, buf_len);
  if (!bounds) wire_bad(st, false, i);
  if (!fpxw::offset_ok(offsets, i + 1, buf_len)) {
    if (i == n - 1) wire_bad(st, false, i);  // the last offset has no message of its own to blame
    bounds = false;
  }
  fpxw::Msg m;
  if (bounds) {
    fpxw::Reader r{buf + offsets[i], buf + offsets[i + 1]};
    const bool ok = WHICH == 0 ? fpxw::parse_proxy_leader_inbound(buf, r, &m) : fpxw::parse_acceptor_inbound(buf, r, &m);
    if (!ok) wire_bad(st, true, i);
  }
  o.kind[i] = m.kind, o.slot[i] = m.slot, o.round[i] = m.round;
  if (o.is_noop) o.is_noop[i] = m.is_noop;
  if (o.value_off) o.value_off[i] = m.value_off;
  if (o.value_len) o.value_len[i] = m.value_len;
  if (o.a) o.a[i] = m.a;
  if (o.b) o.b[i] = m.b;
  if (o.value_id) o.value_id[i] = m.kind == 1 ? o.value_id_base + i : -1;
}

// one thread: a bad message becomes the context's sticky FPX_EINVAL, and every later _dev call of the run applies nothing
__global__ void k_wire_tail(const State st) {
  const int32_t w = st.status[ST_WIRE];
  if (w == 0) return;
  st.status[ST_WIRE] = 0;
  report_abort(st, 1 /*FPX_EINVAL*/, (0x7fffffff - w) & 0x3fffffff, -1, -1);
}

}  // namespace fpx
