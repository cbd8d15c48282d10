"""The JNI shim (frankenpaxos_amd/jni/fpx_jni.c) cannot meet a real JVM here (no JDK in the image): it is
compiled against tests/jni_stub/jni.h and RUN on tests/jni_stub/mock_jvm.c -- a mock of the ten JNIEnv
functions it uses, with bounds-checked "Java arrays" -- driven from python through ctypes.  CPU: the Scala
declarations and the shim agree, every array is length-checked before native code touches it, no
critical regions.  GPU: a fused tick, the wire decoder and the noop ranges through the natives equal the C ABI."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI = os.path.join(ROOT, "frankenpaxos_amd", "jni")
STUB = os.path.join(ROOT, "tests", "jni_stub")
OUT = os.path.join(ROOT, "tests", "build", "libfpxjni_mock.so")


def build_shim():
    import frankenpaxos_amd

    if not os.path.exists(frankenpaxos_amd._lib.SO_PATH):
        frankenpaxos_amd.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    csrc = os.path.join(ROOT, "frankenpaxos_amd", "csrc")
    cmd = ["gcc", "-std=c11", "-O1", "-fPIC", "-shared", "-Wall", "-Werror", "-Wno-unused-parameter", "-I" + STUB,
           os.path.join(JNI, "fpx_jni.c"), os.path.join(STUB, "mock_jvm.c"), "-o", OUT, "-L" + csrc, "-lfpx",
           "-Wl,-rpath," + csrc]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return OUT


@pytest.fixture(scope="module")
def jvm():
    import frankenpaxos_amd

    frankenpaxos_amd.lib()            # libfpx.so (and the HIP runtime it needs) first
    L = C.CDLL(build_shim())
    L.mock_env.restype = C.c_void_p
    L.mock_new_array.restype = C.c_void_p
    L.mock_new_array.argtypes = [C.c_int, C.c_int64, C.c_void_p]
    L.mock_new_direct.restype = C.c_void_p
    L.mock_new_direct.argtypes = [C.c_int64]
    L.mock_data.restype = C.c_void_p
    L.mock_data.argtypes = [C.c_void_p]
    L.mock_free.argtypes = [C.c_void_p]

    class J:
        lib = L
        env = C.c_void_p(L.mock_env())

        @staticmethod
        def arr(a):
            """numpy array -> mock Java array (int[] / long[] / byte[])"""
            a = np.ascontiguousarray(a)
            kind = a.dtype.itemsize
            assert kind in (1, 4, 8)
            return C.c_void_p(L.mock_new_array(kind, a.size, a.ctypes.data))

        @staticmethod
        def read(o, dtype, n):
            return np.ctypeslib.as_array(C.cast(L.mock_data(o), C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), (n,)).copy()

        @staticmethod
        def call(name, restype, *args):
            fn = getattr(L, "Java_frankenpaxos_gpu_Native_" + name)
            fn.restype = restype
            conv = []
            for a in args:
                if isinstance(a, (int, np.integer)):
                    conv.append(C.c_int64(int(a)) if abs(int(a)) > 2 ** 31 else C.c_int32(int(a)))
                else:
                    conv.append(a)
            return fn(J.env, None, *conv)

    return J


def test_scala_natives_have_shim_functions_with_the_same_arity():
    scala = open(os.path.join(JNI, "Native.scala")).read()
    shim = open(os.path.join(JNI, "fpx_jni.c")).read()
    natives = re.findall(r"@native def (\w+)\(([^)]*)\)", scala, flags=re.S)
    assert len(natives) >= 24
    for name, params in natives:
        m = re.search(r"Java_frankenpaxos_gpu_Native_" + name + r"\(\s*JNIEnv\* env, jclass cls,?([^)]*)\)", shim, flags=re.S)
        assert m, name
        n_scala = len([p for p in params.split(",") if p.strip()])
        n_c = len([p for p in m.group(1).split(",") if p.strip()])
        assert n_scala == n_c, (name, n_scala, n_c)
    # ... and no shim function without a Scala declaration
    for name in re.findall(r"Java_frankenpaxos_gpu_Native_(\w+)\(", shim):
        assert name in [n for n, _ in natives], name


def test_no_critical_regions_around_blocking_calls():
    """libfpx entry points allocate, copy over PCIe and synchronise; none of that may run between
    GetPrimitiveArrayCritical and its Release (the JNI specification forbids blocking there, and the GC stalls)"""
    shim = re.sub(r"/\*.*?\*/", "", open(os.path.join(JNI, "fpx_jni.c")).read(), flags=re.S)
    assert "PrimitiveArrayCritical" not in shim


def test_short_arrays_are_rejected_before_native_code_touches_them(jvm):
    """every array shorter than the batch needs is FPX_EINVAL -- with the mock JVM aborting on any out-of-bounds
    region access, reaching the end of this test means the shim never caused one.  No GPU involved: the checks
    come first (handle 0 would be an error of its own)."""
    n = 8
    i32 = lambda k: jvm.arr(np.zeros(k, np.int32))
    i64 = lambda k: jvm.arr(np.zeros(k, np.int64))
    i8 = lambda k: jvm.arr(np.zeros(k, np.int8))
    EINVAL = 1
    ok3 = (i32(n), i32(n), i32(n))
    assert jvm.call("phase2Fused", C.c_int32, 0, n, i32(n - 1), i32(n), i32(n), None, i8(n), i32(n), i32(n), i32(n)) == EINVAL
    assert jvm.call("phase2Fused", C.c_int32, 0, n, *ok3, i64(4 * n - 1), i8(n), i32(n), i32(n), i32(n)) == EINVAL
    assert jvm.call("phase2Fused", C.c_int32, 0, n, *ok3, None, i8(n - 1), i32(n), i32(n), i32(n)) == EINVAL
    assert jvm.call("phase2Fused", C.c_int32, 0, n, *ok3, None, i8(n), i32(n), i32(n), i32(n - 1)) == EINVAL
    assert jvm.call("phase2Fused", C.c_int32, 0, -1, *ok3, None, None, None, None, None) == EINVAL
    assert jvm.call("phase2Fused", C.c_int32, 0, n, None, i32(n), i32(n), None, None, None, None, None) == EINVAL
    assert jvm.call("acceptorPhase2a", C.c_int32, 0, n, *ok3, None, i64(4 * n - 1), None, None) == EINVAL
    assert jvm.call("proxyPhase2b", C.c_int32, 0, n, i32(n), i32(n), i64(4 * n - 4), None, None, None) == EINVAL
    assert jvm.call("proxyOpen", C.c_int32, 0, n, i32(n), i32(n), i32(2), None) == EINVAL
    assert jvm.call("create", C.c_int64, None) == -EINVAL
    assert jvm.call("create", C.c_int64, i32(14)) == -EINVAL            # 15 config fields
    assert jvm.call("quorumEval", C.c_int32, i32(3), 1, i64(4), 1, i8(1)) == EINVAL
    assert jvm.call("acceptorPhase1a", C.c_int32, 0, 0, 0, 0, i64(3), i64(8)) == EINVAL
    assert jvm.call("acceptorPhase1a", C.c_int32, 0, 0, 0, 0, None, i64(7)) == EINVAL
    assert jvm.call("leaderPhase1bScan", C.c_int32, 0, 0, 2, i64(7), 4, i32(1), i32(4), i32(4)) == EINVAL
    assert jvm.call("replicaChosen", C.c_int32, 0, n, i32(n), i32(n - 1), None, i32(2)) == EINVAL
    assert jvm.call("noopRangesFused", C.c_int32, 0, n, 2, i32(n), i32(n), i32(n), None, i64(8 * n - 1), None, None, None, None) == EINVAL
    assert jvm.call("proxyPhase2bNoopRange", C.c_int32, 0, 0, 4, 0, 2, i64(7), i8(1)) == EINVAL
    assert jvm.call("epxPreaccept", C.c_int32, 0, n, 5, i32(n), i32(n), i32(n), i8(n), i8(n), None, i32(5 * n - 1), None, None, None, None) == EINVAL
    assert jvm.call("epxPrepare", C.c_int32, 0, n, 5, i32(n), i32(n), i32(n), i32(n), i8(n), i8(3 * n - 1), None, None) == EINVAL
    assert jvm.call("epxPrepare", C.c_int32, 0, n, 5, i32(n), i32(n), i32(n), i32(n), i8(n), None, None, i32(15 * n - 1)) == EINVAL
    assert jvm.call("epxAccept", C.c_int32, 0, n, i32(n), i32(n), i32(n), i32(n), i32(n - 1), i32(n), i8(n), i8(n), None, None) == EINVAL
    assert jvm.call("epxAccept", C.c_int32, 0, n, i32(n), i32(n), i32(n), i32(n), i32(n), i32(n), i8(n - 1), i8(n), None, None) == EINVAL
    assert jvm.call("epxAccept", C.c_int32, 0, n, i32(n), i32(n), i32(n), i32(n), i32(n), i32(n), i8(n), i8(n), i8(4 * n - 1), None) == EINVAL
    hp = lambda **kw: jvm.call("epxHandlePreaccept", C.c_int32, 0, n, 5, i32(n), i32(n), i32(n), i32(n), i32(n), i8(n),
                               kw.get("tr"), kw.get("din", i32(5 * n)), kw.get("dend"), i8(n), kw.get("rep"), None,
                               kw.get("rd"), kw.get("re"))
    assert hp(din=i32(5 * n - 1)) == EINVAL and hp(dend=i32(n - 1)) == EINVAL and hp(rep=i8(4 * n - 1)) == EINVAL
    assert hp(rd=i32(25 * n - 1)) == EINVAL and hp(re=i32(10 * n - 1)) == EINVAL and hp(tr=i32(n - 1)) == EINVAL
    assert jvm.call("epxReadCmdlog", C.c_int32, 0, 5, 0, 0, 0, i32(10)) == EINVAL      # 5 + n + 1 ints
    assert jvm.call("commUniqueId", C.c_int32, i8(127)) == EINVAL
    assert jvm.call("commCreate", C.c_int32, 0, i8(64), 0, 1) == EINVAL
    assert jvm.call("roundLeader", C.c_int32, 0, 5) == -EINVAL and jvm.call("roundLeader", C.c_int32, 3, 5) == 2
    small = C.c_void_p(jvm.lib.mock_new_direct(4 * n - 1))
    big = lambda: C.c_void_p(jvm.lib.mock_new_direct(4 * n))
    assert jvm.call("phase2FusedDirect", C.c_int32, 0, n, big(), big(), small, None, None, None, None, None) == EINVAL
    assert jvm.call("phase2FusedDirect", C.c_int32, 0, n, big(), big(), i32(n), None, None, None, None, None) == EINVAL  # not direct


def test_wire_decoder_through_the_shim(jvm):
    from frankenpaxos_amd import wire

    msgs = [wire.encode_proxy_leader_phase2a(5, 1, None), wire.encode_proxy_leader_phase2b(0, 2, 300, 1),
            wire.encode_proxy_leader_phase2a(9, 2, bytes.fromhex("0a03616263"))]
    buf, off = wire.pack(msgs)
    d = C.c_void_p(jvm.lib.mock_new_direct(len(buf)))
    C.memmove(jvm.lib.mock_data(d), buf.ctypes.data, len(buf))
    n = len(msgs)
    fields, voff, bad = jvm.arr(np.zeros(7 * n, np.int32)), jvm.arr(np.zeros(n, np.int64)), jvm.arr(np.zeros(1, np.int32))
    assert jvm.call("wireDecodeProxyLeaderInbound", C.c_int32, d, jvm.arr(off), n, fields, voff, bad) == 0
    f = jvm.read(fields, np.int32, 7 * n).reshape(7, n)
    assert f[0].tolist() == [wire.PHASE2A, wire.PHASE2B, wire.PHASE2A]
    assert f[1].tolist() == [5, 300, 9] and f[2].tolist() == [1, 1, 2] and f[3].tolist() == [1, -1, 0]
    assert f[5].tolist() == [-1, 0, -1] and f[6].tolist() == [-1, 2, -1]
    vo = jvm.read(voff, np.int64, n)
    assert bytes(buf[vo[2]:vo[2] + f[4][2]]).hex() == "0a03616263"
    # offsets that run past the direct buffer's capacity: FPX_EINVAL, nothing read
    off2 = off.copy()
    off2[-1] += 100
    assert jvm.call("wireDecodeProxyLeaderInbound", C.c_int32, d, jvm.arr(off2), n, fields, voff, bad) == 1


@pytest.mark.gpu
def test_fused_tick_and_ranges_through_the_shim(jvm, oracle):
    from tests import workloads as W

    S, R, n = 4096, 5, 1000
    cfg = np.array([S, R, 1, 1, 2, 0, 0, 0, 2, 0, 8, 0, 0, 0, 0], np.int32)   # the 15 fpx_config fields
    h = jvm.call("create", C.c_int64, jvm.arr(cfg))
    assert h > 0
    ref = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, f=2, tally_ways=8))
    rng = np.random.default_rng(4)
    slot = rng.permutation(S)[:n].astype(np.int32)
    rnd = np.zeros(n, np.int32)
    val = rng.integers(0, 1 << 30, n).astype(np.int32)
    tgt = W.bits_from_bool(W.random_subsets(rng, n, R, 1, R))
    ch, cr, cv, nr = (jvm.arr(np.zeros(n, t)) for t in (np.int8, np.int32, np.int32, np.int32))
    st = jvm.call("phase2Fused", C.c_int32, h, n, jvm.arr(slot), jvm.arr(rnd), jvm.arr(val), jvm.arr(tgt.view(np.int64)),
                  ch, cr, cv, nr)
    st_r, ch_r, cr_r, cv_r, nr_r = ref.phase2_fused(slot, rnd, val, tgt)
    assert st == st_r == 0
    np.testing.assert_array_equal(jvm.read(ch, np.int8, n), ch_r.astype(np.int8))
    np.testing.assert_array_equal(jvm.read(cv, np.int32, n), cv_r)
    np.testing.assert_array_equal(jvm.read(nr, np.int32, n), nr_r)
    # the same tick again through direct buffers: every (slot, round) is known now, nothing is chosen twice
    def dbuf(a):
        a = np.ascontiguousarray(a)
        d = C.c_void_p(jvm.lib.mock_new_direct(a.nbytes))
        C.memmove(jvm.lib.mock_data(d), a.ctypes.data, a.nbytes)
        return d
    dch = dbuf(np.ones(n, np.int8))
    st = jvm.call("phase2FusedDirect", C.c_int32, h, n, dbuf(slot), dbuf(rnd), dbuf(val), None, dch, None, None, None)
    assert st == 0 and not np.ctypeslib.as_array(C.cast(jvm.lib.mock_data(dch), C.POINTER(C.c_int8)), (n,)).any()
    state = jvm.arr(np.zeros(2, np.int32))
    chosen = ch_r.astype(bool)
    assert jvm.call("replicaChosen", C.c_int32, h, int(chosen.sum()), jvm.arr(slot[chosen]), jvm.arr(cv_r[chosen]), None, state) == 0
    assert tuple(jvm.read(state, np.int32, 2)) == ref.replica_chosen(slot[chosen], cv_r[chosen])[1:]
    assert jvm.call("destroy", C.c_int32, h) == 0
    # Mencius ranges, batched, through the shim
    cfg = np.array([1 << 14, 3, 2, 4, 1, 0, 0, 0, 2, 0, 4, 0, 0, 0, 0], np.int32)
    h = jvm.call("create", C.c_int64, jvm.arr(cfg))
    ref = oracle.System(oracle.make_config(num_slots=1 << 14, num_replicas=3, num_groups=2, num_leader_groups=4, f=1))
    start = np.array([0, 1, 2, 403, 0], np.int32)
    end = np.array([400, 801, 2, 1203, 400], np.int32)
    rr = np.zeros(5, np.int32)
    vb, nw, rch = jvm.arr(np.zeros(5 * 2 * 4, np.int64)), jvm.arr(np.zeros(5, np.int8)), jvm.arr(np.zeros(5, np.int8))
    st = jvm.call("noopRangesFused", C.c_int32, h, 5, 2, jvm.arr(start), jvm.arr(end), jvm.arr(rr), None, vb, None, None, nw, rch)
    b = ref.noop_ranges_fused(start, end, rr)
    assert st == b[0] == 0
    np.testing.assert_array_equal(jvm.read(vb, np.int64, 40).view(np.uint64).reshape(5, 2, 4), b[1])
    np.testing.assert_array_equal(jvm.read(nw, np.int8, 5), b[4].astype(np.int8))
    np.testing.assert_array_equal(jvm.read(rch, np.int8, 5), b[5].astype(np.int8))
    assert jvm.call("destroy", C.c_int32, h) == 0


@pytest.mark.gpu
def test_epaxos_command_log_through_the_shim(jvm, oracle):
    """epxCreateWithLog / epxHandlePreaccept / epxPrepare / epxAccept / epxReadCmdlog on the mock JVM, against the
    oracle: a PreAccept processed at three replicas, the same one again (re-sent replies), a Prepare that moves the
    ballots, the stale PreAccept Nacked, an Accept that commits"""
    n, NI = 5, 32
    h = jvm.call("epxCreateWithLog", C.c_int64, n, 4, 0, NI)
    assert h > 0
    ref = oracle.EPaxos(n, 4, num_instances=NI)
    i32 = lambda a: jvm.arr(np.asarray(a, np.int32))
    i8 = lambda a: jvm.arr(np.asarray(a, np.int8))

    def hp(bo, br, tgt, tr, din):
        rep, nb = jvm.arr(np.zeros(4, np.int8)), jvm.arr(np.zeros(1, np.int32))
        rd, ret = jvm.arr(np.zeros(n * n, np.int32)), jvm.arr(np.zeros(2 * n, np.int32))
        st = jvm.call("epxHandlePreaccept", C.c_int32, h, 1, n, i32([1]), i32([3]), i32([bo]), i32([br]), i32([2]), i8([1]),
                      i32([tr]), i32(din), None, i8([tgt]), rep, nb, rd, ret)
        want = ref.handle_preaccept([1], [3], [bo], [br], [2], [1], [tr], [din], None, [tgt])
        assert st == want[0] == 0
        bits = jvm.read(rep, np.int8, 4).view(np.uint8)
        assert bits.tolist() == [int(want[k][0]) for k in (1, 2, 3, 4)]
        assert int(jvm.read(nb, np.int32, 1)[0]) == int(want[5][0])
        np.testing.assert_array_equal(jvm.read(rd, np.int32, n * n).reshape(n, n), want[6][0])
        got = jvm.read(ret, np.int32, 2 * n)
        np.testing.assert_array_equal(got[:n], want[7][0])
        np.testing.assert_array_equal(got[n:], want[8][0])
        return bits

    assert hp(0, 1, 0b01101, 70, [0, 1, 0, 4, 0]).tolist() == [0b01101, 0, 0, 0]
    assert hp(0, 1, 0b01111, 70, [0, 1, 0, 4, 0]).tolist() == [0b00010, 0b01101, 0, 0]
    rep, nb, po = jvm.arr(np.zeros(3, np.int8)), jvm.arr(np.zeros(1, np.int32)), jvm.arr(np.zeros(3 * n, np.int32))
    assert jvm.call("epxPrepare", C.c_int32, h, 1, n, i32([1]), i32([3]), i32([2]), i32([4]), i8([0b00100]), rep, nb, po) == 0
    want = ref.prepare([1], [3], [2], [4], [0b00100])
    assert jvm.read(rep, np.int8, 3).tolist() == [int(want[1][0]), int(want[2][0]), int(want[3][0])]
    np.testing.assert_array_equal(jvm.read(po, np.int32, 3 * n), np.concatenate([want[5][0], want[6][0], want[7][0]]))
    assert hp(0, 1, 0b00100, 70, [0, 1, 0, 4, 0]).tolist() == [0, 0, 0b00100, 0]
    rep4, nb = jvm.arr(np.zeros(4, np.int8)), jvm.arr(np.zeros(1, np.int32))
    # a handle is the authority on its own n: a caller that says otherwise is refused (it would size the replies short)
    assert jvm.call("epxPrepare", C.c_int32, h, 1, 3, i32([1]), i32([3]), i32([2]), i32([4]), i8([0b00100]), rep, nb, po) == 1
    assert jvm.call("epxAccept", C.c_int32, h, 1, i32([1]), i32([3]), i32([2]), i32([4]), i32([71]), i32([2]), i8([1]),
                    i8([0b00101]), rep4, nb) == 0
    want = ref.accept([1], [3], [2], [4], [71], [0b00101], [2], [1])
    assert jvm.read(rep4, np.int8, 4).view(np.uint8).tolist() == [int(want[k][0]) for k in (1, 2, 3, 5)]
    entry = jvm.arr(np.zeros(6 + n, np.int32))
    for r in range(n):
        assert jvm.call("epxReadCmdlog", C.c_int32, h, n, r, 1, 3, entry) == 0
        got = jvm.read(entry, np.int32, 6 + n)
        assert tuple(got[:5]) == ref.read_cmdlog(r, 1, 3)
        deps, end = ref.read_cmdlog_deps(r, 1, 3)
        assert got[5:5 + n].tolist() == deps.tolist() and got[5 + n] == end
    # a Commit from a replica outside (GpuEPaxosReplica's Request.Commit branch -> Native.epxHandleCommit): recorded at ONE
    # replica, with its dependencies; then by triple id alone at another; a short array is refused
    assert jvm.call("epxHandleCommit", C.c_int32, h, 2, n, i32([2, 0]), i32([7, 1]), i32([90, 91]), i32([3, -1]), i8([0, 0]),
                    i32([1, 0, 5, 0, 2, 0, 0, 0, 0, 0]), i32([10, 0]), i8([0b01000, 0b00011])) == 0
    assert ref.handle_commit([2, 0], [7, 1], [90, 91], [0b01000, 0b00011], key=[3, -1], is_set=[0, 0],
                             deps=[[1, 0, 5, 0, 2], [0, 0, 0, 0, 0]], deps_values_end=[10, 0]) == 0
    assert jvm.call("epxHandleCommit", C.c_int32, h, 1, n, i32([3]), i32([2]), i32([92]), i32([1]), i8([1]), None, None, i8([0b10000])) == 0
    assert ref.handle_commit([3], [2], [92], [0b10000], key=[1], is_set=[1]) == 0
    for r, L, x in ((3, 2, 7), (0, 0, 1), (1, 0, 1), (4, 3, 2), (2, 2, 7)):
        assert jvm.call("epxReadCmdlog", C.c_int32, h, n, r, L, x, entry) == 0
        got = jvm.read(entry, np.int32, 6 + n)
        assert tuple(got[:5]) == ref.read_cmdlog(r, L, x)
        deps, end = ref.read_cmdlog_deps(r, L, x)
        assert got[5:5 + n].tolist() == deps.tolist() and got[5 + n] == end
    assert jvm.call("epxHandleCommit", C.c_int32, h, 2, n, i32([3]), i32([2]), i32([92]), i32([1]), i8([1]), None, None, i8([0b10000])) == 1
    assert jvm.call("epxDestroy", C.c_int32, h) == 0


@pytest.mark.gpu
def test_a_leader_change_on_wire_bytes_through_the_shim(jvm, oracle):
    """The sequence the unchanged Leader drives (multipaxos/Leader.scala:231, 410-420, 504-577, 672-697), on wire
    bytes, through the natives GpuAcceptor / GpuProxyLeader call (frankenpaxos_amd/jni/Native.scala):
      1. leader 0: Phase1a(round 0) to every acceptor address -> Phase1b with no votes
      2. a tick of Phase2a (round 0, slots 0..199; slots >= 100 reach acceptors 0 and 1 only) -> Chosen
      3. leader 1 takes over: Phase1a(round 1, chosenWatermark 50) reaches acceptors 0 and 1 -> Phase1b with their votes
      4. leader 0, unaware, sends a tick in round 0 (slots 200..209): acceptors 0 and 1 Nack with round 1, acceptor 2
         still votes -- nothing is chosen, Nack(1) goes to leaders(roundSystem.leader(0))
      5. leader 0 reacts (Leader.handleNack): Phase1a(nextClassicRound = 2) to all three; the Phase1b of acceptor 2
         carries its round-0 votes of step 4, the safe values (Leader.scala:306-329) make leader 0 re-propose them
      6. the re-proposals in round 2 are chosen
    Every integer that comes out equals the oracle fed the same calls; every byte string equals the python codec
    (pinned on google.protobuf, tests/test_wire.py)."""
    from frankenpaxos_amd import wire

    S, R = 4096, 3
    cfg = np.array([S, R, 1, 1, 1, 0, 0, 0, 2, 0, 8, 0, 0, 0, 0], np.int32)   # f = 1, 2 leaders, acceptor-scalar rounds
    h = jvm.call("create", C.c_int64, jvm.arr(cfg))
    assert h > 0
    ref = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, f=1, tally_ways=8, num_leaders=2))
    i32 = lambda a: jvm.arr(np.asarray(a, np.int32))
    payload = {}                                     # value id -> the CommandBatchOrNoop bytes the JVM keeps

    def direct(b):
        d = C.c_void_p(jvm.lib.mock_new_direct(max(1, len(b))))
        if len(b):
            C.memmove(jvm.lib.mock_data(d), bytes(b), len(b))
        return d

    def dbytes(d, n):
        return bytes(np.ctypeslib.as_array(C.cast(jvm.lib.mock_data(d), C.POINTER(C.c_uint8)), (n,)))

    def phase1a_at(acceptor, msg):
        """AcceptorInbound bytes at one acceptor address -> the LeaderInbound bytes it answers with"""
        buf, off = wire.pack([msg])
        fields, bad = jvm.arr(np.zeros(6, np.int32)), jvm.arr(np.zeros(1, np.int32))
        assert jvm.call("wireDecodeAcceptorInbound", C.c_int32, direct(buf), jvm.arr(off), 1, fields, None, bad) == 0
        kind, _, rnd, _, _, wm = jvm.read(fields, np.int32, 6).tolist()
        assert kind == wire.PHASE1A
        bits = jvm.arr(np.zeros(8, np.int64))
        tgt = oracle.bits_of([acceptor])
        assert jvm.call("acceptorPhase1a", C.c_int32, h, 0, rnd, wm, jvm.arr(tgt.view(np.int64)), bits) == 0
        st, pb, nb = ref.acceptor_phase1a(0, rnd, wm, tgt)
        got = jvm.read(bits, np.int64, 8).view(np.uint64)
        assert st == 0 and got[:4].tolist() == pb.tolist() and got[4:].tolist() == nb.tolist()
        out = direct(b"\0" * 65536)
        if nb.any():                                  # Acceptor.scala:155-162
            cur = jvm.call("acceptorRound", C.c_int32, h, 0, acceptor)
            assert cur == ref.read_acceptor(0, acceptor)[0]
            n = jvm.call("wireEncodeLeaderNack", C.c_int64, out, cur)
            assert dbytes(out, n) == wire.encode_leader_nack(cur)
            return dbytes(out, n)
        cap = 1024                                    # Acceptor.scala:163-181
        sl, vr, vv = (jvm.arr(np.zeros(cap, np.int32)) for _ in range(3))
        k = jvm.call("acceptorPhase1bInfo", C.c_int32, h, 0, acceptor, wm, cap, sl, vr, vv)
        want = ref.acceptor_phase1b_info(0, acceptor, wm)
        assert k == len(want[0])
        sl, vr, vv = (jvm.read(x, np.int32, k) for x in (sl, vr, vv))
        for a, b in zip((sl, vr, vv), want):
            np.testing.assert_array_equal(a, b)
        blobs = [payload[int(v)] for v in vv]
        vbuf, voff = wire.pack(blobs)
        n = jvm.call("wireEncodeLeaderPhase1b", C.c_int64, out, 0, acceptor, rnd, k, i32(sl), i32(vr), direct(vbuf),
                     jvm.arr(voff[:-1] if k else np.zeros(1, np.int64)), i32(np.diff(voff) if k else [0]), None)
        assert n > 0
        assert dbytes(out, n) == wire.encode_leader_phase1b(0, acceptor, rnd, [(int(s), int(r), p) for s, r, p in zip(sl, vr, blobs)])
        return dbytes(out, n)

    def tick(msgs, target_bits):
        """ProxyLeaderInbound{Phase2a} bytes -> (chosen ReplicaInbound bytes, Nack LeaderInbound bytes with their leader)"""
        buf, off = wire.pack(msgs)
        n = len(msgs)
        fields, voff, bad = jvm.arr(np.zeros(7 * n, np.int32)), jvm.arr(np.zeros(n, np.int64)), jvm.arr(np.zeros(1, np.int32))
        assert jvm.call("wireDecodeProxyLeaderInbound", C.c_int32, direct(buf), jvm.arr(off), n, fields, voff, bad) == 0
        f = jvm.read(fields, np.int32, 7 * n).reshape(7, n)
        vo = jvm.read(voff, np.int64, n)
        assert (f[0] == wire.PHASE2A).all()
        slot, rnd = f[1].copy(), f[2].copy()
        val = np.zeros(n, np.int32)
        for i in range(n):                            # intern: value id = next free index (GpuPhase2Engine.intern)
            val[i] = len(payload)
            payload[int(val[i])] = bytes(buf[vo[i]:vo[i] + f[4][i]])
        tgt = np.tile(target_bits, (n, 1))
        ch, cr, cv, nr = (jvm.arr(np.zeros(n, t)) for t in (np.int8, np.int32, np.int32, np.int32))
        assert jvm.call("phase2Fused", C.c_int32, h, n, i32(slot), i32(rnd), i32(val), jvm.arr(tgt.view(np.int64)), ch, cr, cv, nr) == 0
        st, ch_r, cr_r, cv_r, nr_r = ref.phase2_fused(slot, rnd, val, tgt)
        ch, cv, nr = jvm.read(ch, np.int8, n), jvm.read(cv, np.int32, n), jvm.read(nr, np.int32, n)
        assert st == 0
        np.testing.assert_array_equal(ch, ch_r.astype(np.int8))
        np.testing.assert_array_equal(cv, cv_r)
        np.testing.assert_array_equal(nr, nr_r)
        chosen = [wire.encode_replica_chosen(int(slot[i]), payload[int(cv[i])]) for i in range(n) if ch[i]]
        nacks = [(jvm.call("roundLeader", C.c_int32, 2, int(rnd[i])), wire.encode_leader_nack(int(nr[i]))) for i in range(n) if nr[i] >= 0]
        return chosen, nacks

    cmd = lambda s, tag: bytes.fromhex("0a") + bytes([len(b"%s %d" % (tag, s)) + 0]) + b"%s %d" % (tag, s)  # opaque CommandBatch bytes
    everyone, zero_one = oracle.bits_of([0, 1, 2]), oracle.bits_of([0, 1])
    # 1
    for a in range(R):
        reply = phase1a_at(a, wire.encode_acceptor_phase1a(0, 0))
        assert reply.hex() == "0a06" + "0800" + "10%02x" % a + "1800"          # Phase1b(group 0, acceptor a, round 0, no info)
    # 2
    chosen, nacks = tick([wire.encode_proxy_leader_phase2a(s, 0, cmd(s, b"set")) for s in range(100)], everyone)
    assert len(chosen) == 100 and not nacks
    chosen, nacks = tick([wire.encode_proxy_leader_phase2a(s, 0, cmd(s, b"set")) for s in range(100, 200)], zero_one)
    assert len(chosen) == 100 and not nacks
    d = wire.decode_replica_inbound(chosen)
    assert d["slot"].tolist() == list(range(100, 200))
    assert bytes(d["buf"][d["value_off"][7]:d["value_off"][7] + d["value_len"][7]]) == cmd(107, b"set")
    # 3
    for a in (0, 1):
        d = wire.decode_leader_inbound([phase1a_at(a, wire.encode_acceptor_phase1a(1, 50))])
        assert d["kind"].tolist() == [wire.PHASE1B] and d["round"].tolist() == [1] and d["acceptor_index"].tolist() == [a]
        assert d["info_slot"].tolist() == list(range(50, 200)) and set(d["info_vote_round"].tolist()) == {0}
        assert bytes(d["buf"][d["info_value_off"][0]:d["info_value_off"][0] + d["info_value_len"][0]]) == cmd(50, b"set")
    # 4
    chosen, nacks = tick([wire.encode_proxy_leader_phase2a(s, 0, cmd(s, b"stale")) for s in range(200, 210)], everyone)
    assert not chosen and nacks == [(0, bytes.fromhex("32020801"))] * 10      # Nack(round 1) to leader 0
    # 5
    nxt = jvm.call("roundLeader", C.c_int32, 2, 2)
    assert nxt == 0                                                            # round 2 is leader 0's (ClassicRoundRobin)
    infos = [wire.decode_leader_inbound([phase1a_at(a, wire.encode_acceptor_phase1a(2, 50))]) for a in range(R)]
    assert infos[0]["info_slot"].tolist() == list(range(50, 200)) and infos[2]["info_slot"].tolist() == list(range(50, 100)) + list(range(200, 210))
    mx, sr, sv = (jvm.arr(np.zeros(k, np.int32)) for k in (1, 256, 256))
    assert jvm.call("leaderPhase1bScan", C.c_int32, h, 50, 1, jvm.arr(everyone.view(np.int64)), 256, mx, sr, sv) == 0
    st, mx_r, sr_r, sv_r = ref.leader_phase1b_scan(50, everyone, 256)
    assert st == 0 and int(jvm.read(mx, np.int32, 1)[0]) == mx_r == 209
    k = mx_r - 50 + 1
    safe_round, safe_value = jvm.read(sr, np.int32, k), jvm.read(sv, np.int32, k)
    np.testing.assert_array_equal(safe_round, sr_r)
    np.testing.assert_array_equal(safe_value, sv_r)
    assert (safe_round == 0).all() and payload[int(safe_value[-1])] == cmd(209, b"stale")
    # 6  the safe values go out again in round 2 (Leader.scala:549-566)
    chosen, nacks = tick([wire.encode_proxy_leader_phase2a(50 + j, 2, payload[int(safe_value[j])]) for j in range(k)], everyone)
    assert len(chosen) == k and not nacks
    d = wire.decode_replica_inbound(chosen)
    assert d["slot"].tolist() == list(range(50, 210))
    assert bytes(d["buf"][d["value_off"][-1]:d["value_off"][-1] + d["value_len"][-1]]) == cmd(209, b"stale")
    # the log window: the first 2048 rows are recycled on both sides, the states stay equal, the rows vote afresh
    assert jvm.call("recycleSlots", C.c_int32, h, 0, 2048) == 0
    ref.recycle_slots(0, 2048)
    cap = 8
    sl, vr, vv = (jvm.arr(np.zeros(cap, np.int32)) for _ in range(3))
    assert jvm.call("acceptorPhase1bInfo", C.c_int32, h, 0, 0, 0, cap, sl, vr, vv) == 0 == len(ref.acceptor_phase1b_info(0, 0, 0)[0])
    chosen, nacks = tick([wire.encode_proxy_leader_phase2a(s, 2, cmd(s, b"lap2")) for s in range(0, 64)], everyone)
    assert len(chosen) == 64 and not nacks
    # a short array is refused before native code touches it
    assert jvm.call("acceptorPhase1bInfo", C.c_int32, h, 0, 0, 0, 16, sl, vr, vv) == -1
    assert jvm.call("destroy", C.c_int32, h) == 0


@pytest.mark.gpu
def test_the_acceptors_read_path_through_the_shim(jvm, oracle):
    """GpuAcceptor answers MaxSlotRequest / BatchMaxSlotRequest with Acceptor.maxVotedSlot (multipaxos/Acceptor.scala:
    222-254) as GpuPhase2Engine follows it (jni/Native.scala: noteVotes / maxVotedSlot), on wire bytes through the
    natives it calls:
      1. thrifty ticks nobody Nacks: every targeted acceptor voted -- the engine's bookkeeping alone == the oracle's
         Acceptor.maxVotedSlot, no native asked
      2. a competing leader pre-promises acceptor 1; the old leader's tick is Nacked by it: the group is stale, the
         read asks the device (acceptorMaxVotedIn, lap by lap) and is exact again: acceptor 1 did NOT vote, 0 and 2 did
      3. the window wraps (rows 0..2047 recycled, slots 4096.. land in them): the device's scalar over ROWS is no longer
         the maximum over SLOTS; the two-lap read is
      4. the requests and replies as bytes: AcceptorInbound{MaxSlotRequest}, {BatchMaxSlotRequest} decoded by the shim's
         decoder, ClientInbound{MaxSlotReply} / ReadBatcherInbound{BatchMaxSlotReply} carrying that slot"""
    from frankenpaxos_amd import wire

    S, R, F = 4096, 3, 1
    cfg = np.array([S, R, 1, 1, F, 0, 0, 0, 2, 0, 8, 0, 0, 0, 0], np.int32)
    h = jvm.call("create", C.c_int64, jvm.arr(cfg))
    assert h > 0
    ref = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, f=F, tally_ways=8, num_leaders=2))
    i32 = lambda a: jvm.arr(np.asarray(a, np.int32))
    base = 0                                                   # GpuPhase2Engine.base: first slot of the window
    max_voted, stale, voted_truth = [-1] * R, [False], [-1] * R  # the engine's table; what really happened (by hand)

    def phase1a(rnd, acceptors):
        bits = jvm.arr(np.zeros(8, np.int64))
        tgt = oracle.bits_of(acceptors)
        assert jvm.call("acceptorPhase1a", C.c_int32, h, 0, rnd, 0, jvm.arr(tgt.view(np.int64)), bits) == 0
        assert ref.acceptor_phase1a(0, rnd, 0, tgt)[0] == 0

    def tick(slots, rnd, targets, check_oracle=True):
        """GpuPhase2Engine.phase2Tick + noteVotes: slots are log positions, rows = slot % S"""
        n = len(slots)
        rows = np.asarray(slots, np.int32) % S
        val = np.arange(n, dtype=np.int32) + 7
        tgt = np.stack([oracle.bits_of(t) for t in targets])
        ch, cr, cv, nr = (jvm.arr(np.zeros(n, t)) for t in (np.int8, np.int32, np.int32, np.int32))
        assert jvm.call("phase2Fused", C.c_int32, h, n, i32(rows), i32(np.full(n, rnd)), i32(val), jvm.arr(tgt.view(np.int64)), ch, cr, cv, nr) == 0
        nr = jvm.read(nr, np.int32, n)
        if check_oracle:
            st, ch_r, cr_r, cv_r, nr_r = ref.phase2_fused(rows, np.full(n, rnd, np.int32), val, tgt)
            assert st == 0
            np.testing.assert_array_equal(nr, nr_r)
        covered = set()
        for i in sorted(range(n), key=lambda i: -int(slots[i])):          # noteVotes
            if nr[i] >= 0:
                stale[0] = True
            else:
                for a in targets[i]:
                    if a not in covered:
                        covered.add(a)
                        max_voted[a] = max(max_voted[a], int(slots[i]))
        return nr

    def read(a):
        """GpuPhase2Engine.maxVotedSlot"""
        if stale[0]:
            r0 = base % S
            for b in range(R):
                hi = jvm.call("acceptorMaxVotedIn", C.c_int32, h, 0, b, 0, r0) if r0 > 0 else -1
                lo = jvm.call("acceptorMaxVotedIn", C.c_int32, h, 0, b, r0, S - r0) if hi < 0 else -1
                assert hi >= -1 and lo >= -1
                slot = base + (S - r0) + hi if hi >= 0 else (base + (lo - r0) if lo >= 0 else -1)
                max_voted[b] = max(max_voted[b], slot)
            stale[0] = False
        return max_voted[a]

    # 1. rotating windows of f + 1 = 2 neighbours, nobody Nacks
    phase1a(0, [0, 1, 2])
    slots = list(range(0, 300))
    targets = [[(s % R), (s + 1) % R] for s in slots]
    tick(slots, 0, targets)
    for a in range(R):
        voted_truth[a] = max(s for s, t in zip(slots, targets) if a in t)
        assert read(a) == voted_truth[a] == ref.read_acceptor(0, a)[1]
    # 2. acceptor 1 promises round 1; the round-0 tick that follows is Nacked by it wherever it is a target
    phase1a(1, [1])
    slots = list(range(300, 330))
    targets = [[(s % R), (s + 1) % R] for s in slots]
    nr = tick(slots, 0, targets)
    assert (nr >= 0).any() and stale[0]
    for a in (0, 2):
        voted_truth[a] = max(s for s, t in zip(slots, targets) if a in t)
    for a in range(R):
        assert read(a) == voted_truth[a] == ref.read_acceptor(0, a)[1]
    assert not stale[0] and voted_truth[1] < 300                      # acceptor 1 has not voted since its promise
    # 2b. ADVICE r05: a Nacked tick leaves the group stale, and its rows are recycled BEFORE anybody reads -- the votes
    #     acceptors 0 and 2 cast in it must not be lost with the rows (Acceptor.scala:208 never misses a vote, and a low
    #     MaxSlotReply.slot is the unsafe direction): advanceWindow rescans a stale group before it clears rows
    slots = list(range(1000, 1030))
    targets = [[(s % R), (s + 1) % R] for s in slots]
    nr = tick(slots, 0, targets)
    assert (nr >= 0).any() and stale[0]
    for a in (0, 2):
        voted_truth[a] = max(s for s, t in zip(slots, targets) if a in t)
    if stale[0]:                                                       # GpuPhase2Engine.advanceWindow: refreshStaleGroups()
        read(0)
    assert jvm.call("recycleSlots", C.c_int32, h, 0, 2048) == 0
    ref.recycle_slots(0, 2048)
    for a in range(R):
        assert read(a) == voted_truth[a] == ref.read_acceptor(0, a)[1]
    # (the lazy rescan of round 5 -- recycle first, ask the device at the next read -- finds the rows empty:)
    assert jvm.call("acceptorMaxVotedIn", C.c_int32, h, 0, 0, 0, 2048) == -1
    # 3. the new leader takes over (round 1) and proposes far ahead in the window; then the window moves on: rows
    #    0 .. 2047 are recycled, base = 2048, and the log goes on in slots 4096 .. -- in rows 0 ..
    phase1a(1, [0, 2])
    slots = list(range(3000, 3030))
    targets = [[(s % R), (s + 1) % R] for s in slots]
    tick(slots, 1, targets)
    for a in range(R):
        voted_truth[a] = max(s for s, t in zip(slots, targets) if a in t)
        assert read(a) == voted_truth[a] == ref.read_acceptor(0, a)[1]
    assert jvm.call("recycleSlots", C.c_int32, h, 0, 2048) == 0
    base = 2048
    slots = list(range(4096, 4160))
    targets = [[(s % R), (s + 1) % R] for s in slots]
    tick(slots, 1, targets, check_oracle=False)
    for a in range(R):
        voted_truth[a] = max(s for s, t in zip(slots, targets) if a in t)
        assert read(a) == voted_truth[a]
    # ... and if the engine has to ask (a Nack somewhere): the two laps, newest first -- not the scalar over rows
    stale[0], max_voted[:] = True, [-1] * R
    for a in range(R):
        assert read(a) == voted_truth[a]
    # (the maximum over ALL rows -- what the device's scalar is -- is a row of the OLD lap: slot 3029's, not slot 4158's row 62)
    assert jvm.call("acceptorMaxVotedIn", C.c_int32, h, 0, 0, 0, S) == 3029 and voted_truth[0] == 4158
    assert jvm.call("acceptorMaxVotedIn", C.c_int32, h, 0, 0, 0, S + 1) < -1      # FPX_EINVAL comes back as -(status + 1)
    # 4. on the wire
    cid = bytes.fromhex("0a0d") + b"10.0.0.1:9000" + bytes.fromhex("1003" "1811")        # CommandId{address, pseudonym 3, id 17}
    req = bytes([0x1a, len(cid) + 2, 0x0a, len(cid)]) + cid                                  # AcceptorInbound{max_slot_request = 3}
    breq = bytes.fromhex("22" "04" "0805" "1007")                                           # {batch_max_slot_request = 4 {5, 7}}
    buf, off = wire.pack([req, breq])
    fields, voff, bad = jvm.arr(np.zeros(12, np.int32)), jvm.arr(np.zeros(2, np.int64)), jvm.arr(np.zeros(1, np.int32))
    d = C.c_void_p(jvm.lib.mock_new_direct(len(buf)))
    C.memmove(jvm.lib.mock_data(d), bytes(buf), len(buf))
    assert jvm.call("wireDecodeAcceptorInbound", C.c_int32, d, jvm.arr(off), 2, fields, voff, bad) == 0
    f = jvm.read(fields, np.int32, 12).reshape(6, 2)
    vo = jvm.read(voff, np.int64, 2)
    assert f[0].tolist() == [wire.MAX_SLOT_REQUEST, wire.BATCH_MAX_SLOT_REQUEST]
    assert bytes(buf[vo[0]:vo[0] + f[4][0]]) == cid and (f[1][1], f[2][1]) == (5, 7)
    reply = wire.encode_client_max_slot_reply(bytes(buf[vo[0]:vo[0] + f[4][0]]), 0, 2, read(2))
    assert reply == bytes([0x22, len(cid) + 2 + 7]) + bytes([0x0a, len(cid)]) + cid + bytes.fromhex("1000" "1802") + bytes([0x20]) + bytes.fromhex("bf20")
    breply = wire.encode_read_batcher_batch_max_slot_reply(5, 7, 1, read(1))
    assert breply == bytes.fromhex("22" "09" "0805" "1007" "1801" "20bf20")
    assert jvm.call("destroy", C.c_int32, h) == 0


def _direct(jvm, b):
    d = C.c_void_p(jvm.lib.mock_new_direct(max(1, len(b))))
    if len(b):
        C.memmove(jvm.lib.mock_data(d), bytes(b), len(b))
    return d


def _dbytes(jvm, d, n):
    return bytes(np.ctypeslib.as_array(C.cast(jvm.lib.mock_data(d), C.POINTER(C.c_uint8)), (n,)))


@pytest.mark.gpu
def test_a_mencius_leader_change_on_wire_bytes_through_the_shim(jvm, oracle):
    """What the unchanged mencius Leaders drive (mencius/Leader.scala:342-345, 455, 486-491, 288-297), on wire bytes,
    through the natives GpuMenciusAcceptor / GpuMenciusProxyLeader call (frankenpaxos_amd/jni/MenciusNative.scala).
    2 leader groups x 2 acceptor groups x 3 acceptors (f = 1), 2 leaders per group:
      1. the leaders of both groups run Phase 1 in round 0: Phase1a to every acceptor address of their group
      2. a burst: group 0 proposes commands in its slots, group 1 skips its slots with a noop range, then proposes
      3. in group 1 leader 1 takes over: Phase1a(round 1, chosenWatermark 9) reaches two acceptors of each of its
         acceptor groups -> Phase1b with the group's votes from the watermark on (noops of the range included)
      4. the old leader of group 1, unaware, sends a command and a noop range in round 0: the acceptors that promised
         Nack with round 1 (LeaderInbound field 7), nothing is chosen; group 0 is not disturbed
      5. the new leader's re-proposals and a noop range in round 1 are chosen
    Every integer equals the oracle fed the same calls; every byte string equals the python codec (pinned on
    google.protobuf, tests/test_wire.py)."""
    from frankenpaxos_amd import wire

    S, R, A, L = 1024, 3, 2, 2
    cfg = np.array([S, R, A, L, 1, 0, 0, 0, 2, 0, 8, 0, 0, 0, 0], np.int32)
    h = jvm.call("create", C.c_int64, jvm.arr(cfg))
    assert h > 0
    ref = oracle.System(oracle.make_config(num_slots=S, num_replicas=R, num_groups=A, num_leader_groups=L, f=1,
                                           tally_ways=8, num_leaders=2))
    i32 = lambda a: jvm.arr(np.asarray(a, np.int32))
    payload = {}
    group_of = lambda slot: (slot % L) * A + (slot // L) % A       # mencius/ProxyLeader.scala:169-176, 231-234

    def phase1a_at(group, acceptor, msg):
        """mencius AcceptorInbound{Phase1a} bytes at one acceptor address -> the LeaderInbound bytes it answers with"""
        buf, off = wire.pack([msg])
        fields, bad = jvm.arr(np.zeros(7, np.int32)), jvm.arr(np.zeros(1, np.int32))
        assert jvm.call("wireMenciusDecodeAcceptorInbound", C.c_int32, _direct(jvm, buf), jvm.arr(off), 1, fields, None, bad) == 0
        kind, _, _, rnd, _, _, wm = jvm.read(fields, np.int32, 7).tolist()
        assert kind == wire.PHASE1A
        bits = jvm.arr(np.zeros(8, np.int64))
        tgt = oracle.bits_of([acceptor])
        assert jvm.call("acceptorPhase1a", C.c_int32, h, group, rnd, wm, jvm.arr(tgt.view(np.int64)), bits) == 0
        st, pb, nb = ref.acceptor_phase1a(group, rnd, wm, tgt)
        got = jvm.read(bits, np.int64, 8).view(np.uint64)
        assert st == 0 and got[:4].tolist() == pb.tolist() and got[4:].tolist() == nb.tolist()
        out = _direct(jvm, b"\0" * 65536)
        if nb.any():                                  # mencius/Acceptor.scala:173-180
            cur = jvm.call("acceptorRound", C.c_int32, h, group, acceptor)
            assert cur == ref.read_acceptor(group, acceptor)[0]
            n = jvm.call("wireMenciusEncodeLeaderNack", C.c_int64, out, cur)
            assert _dbytes(jvm, out, n) == wire.mencius_encode("leader_nack", cur)
            return _dbytes(jvm, out, n)
        cap = 1024                                    # mencius/Acceptor.scala:184-199
        sl, vr, vv = (jvm.arr(np.zeros(cap, np.int32)) for _ in range(3))
        k = jvm.call("acceptorPhase1bInfo", C.c_int32, h, group, acceptor, wm, cap, sl, vr, vv)
        want = ref.acceptor_phase1b_info(group, acceptor, wm)
        assert k == len(want[0])
        sl, vr, vv = (jvm.read(x, np.int32, k) for x in (sl, vr, vv))
        for a, b in zip((sl, vr, vv), want):
            np.testing.assert_array_equal(a, b)
        blobs = [payload[int(v)] if v >= 0 else b"" for v in vv]
        vbuf, voff = wire.pack(blobs)
        noop = jvm.arr((vv < 0).astype(np.int8)) if k else None
        n = jvm.call("wireEncodeLeaderPhase1b", C.c_int64, out, group % A, acceptor, rnd, k, i32(sl), i32(vr), _direct(jvm, vbuf),
                     jvm.arr(voff[:-1] if k else np.zeros(1, np.int64)), i32(np.diff(voff) if k else [0]), noop)
        assert n > 0
        info = [(int(s), int(r), (None if v < 0 else payload[int(v)])) for s, r, v in zip(sl, vr, vv)]
        assert _dbytes(jvm, out, n) == wire.encode_leader_phase1b(group % A, acceptor, rnd, info)   # (the same layout as MultiPaxos')
        return _dbytes(jvm, out, n)

    def burst(msgs, targets=None):
        """mencius ProxyLeaderInbound bytes (Phase2a and Phase2aNoopRange, in arrival order) -> what the proxy leader sends:
        ReplicaInbound bytes (Chosen / ChosenNoopRange) and (leader group, leader, LeaderInbound{Nack} bytes)"""
        buf, off = wire.pack(msgs)
        n = len(msgs)
        fields, voff, bad = jvm.arr(np.zeros(8 * n, np.int32)), jvm.arr(np.zeros(n, np.int64)), jvm.arr(np.zeros(1, np.int32))
        assert jvm.call("wireMenciusDecodeProxyLeaderInbound", C.c_int32, _direct(jvm, buf), jvm.arr(off), n, fields, voff, bad) == 0
        f = jvm.read(fields, np.int32, 8 * n).reshape(8, n)
        vo = jvm.read(voff, np.int64, n)
        want = wire.mencius_decode_proxy_leader_inbound(msgs)
        for j, name in enumerate(("kind", "slot", "slot_end", "round", "is_noop", "value_len", "group_index", "acceptor_index")):
            np.testing.assert_array_equal(f[j], want[name], err_msg=name)
        sent, nacked = [], []
        # maximal runs of one kind, in arrival order (GpuMenciusProxyLeader.flushTick)
        i = 0
        while i < n:
            j = i
            while j < n and f[0][j] == f[0][i]:
                j += 1
            idx = np.arange(i, j)
            if f[0][i] == wire.PHASE2A:
                slot, rnd = f[1][idx].copy(), f[3][idx].copy()
                val = np.zeros(len(idx), np.int32)
                for k, m in enumerate(idx):
                    if f[4][m]:
                        val[k] = -1
                    else:
                        val[k] = len(payload)
                        payload[int(val[k])] = bytes(buf[vo[m]:vo[m] + f[5][m]])
                tgt = None if targets is None else np.tile(targets, (len(idx), 1))
                ch, cr, cv, nr = (jvm.arr(np.zeros(len(idx), t)) for t in (np.int8, np.int32, np.int32, np.int32))
                assert jvm.call("phase2Fused", C.c_int32, h, len(idx), i32(slot), i32(rnd), i32(val),
                                None if tgt is None else jvm.arr(tgt.view(np.int64)), ch, cr, cv, nr) == 0
                st, ch_r, cr_r, cv_r, nr_r = ref.phase2_fused(slot, rnd, val, tgt)
                ch, cv, nr = jvm.read(ch, np.int8, len(idx)), jvm.read(cv, np.int32, len(idx)), jvm.read(nr, np.int32, len(idx))
                assert st == 0
                np.testing.assert_array_equal(ch, ch_r.astype(np.int8))
                np.testing.assert_array_equal(cv, cv_r)
                np.testing.assert_array_equal(nr, nr_r)
                for k in range(len(idx)):
                    if ch[k]:
                        sent.append(wire.mencius_encode("replica_chosen", int(slot[k]), None if cv[k] < 0 else payload[int(cv[k])]))
                    if nr[k] >= 0:
                        nacked.append((int(slot[k]) % L, jvm.call("roundLeader", C.c_int32, 2, int(rnd[k])), wire.mencius_encode("leader_nack", int(nr[k]))))
            else:
                assert f[0][i] == wire.PHASE2A_NOOP_RANGE
                m = len(idx)
                start, end, rnd = f[1][idx].copy(), f[2][idx].copy(), f[3][idx].copy()
                tm = None if targets is None else np.tile(targets, (m, A, 1))
                votes, nacks = jvm.arr(np.zeros(m * A * 4, np.int64)), jvm.arr(np.zeros(m * A * 4, np.int64))
                nr, isnew, chosen = jvm.arr(np.zeros(m, np.int32)), jvm.arr(np.zeros(m, np.int8)), jvm.arr(np.zeros(m, np.int8))
                assert jvm.call("noopRangesFused", C.c_int32, h, m, A, i32(start), i32(end), i32(rnd),
                                None if tm is None else jvm.arr(tm.reshape(-1).view(np.int64)), votes, nacks, nr, isnew, chosen) == 0
                st, vb_r, nb_r, nr_r, new_r, ch_r = ref.noop_ranges_fused(start, end, rnd, tm)
                assert st == 0
                np.testing.assert_array_equal(jvm.read(votes, np.int64, m * A * 4).view(np.uint64).reshape(m, A, 4), vb_r)
                np.testing.assert_array_equal(jvm.read(nr, np.int32, m), nr_r)
                ch = jvm.read(chosen, np.int8, m)
                np.testing.assert_array_equal(ch, ch_r.astype(np.int8))
                for k in range(m):
                    if ch[k]:
                        sent.append(wire.mencius_encode("replica_chosen_noop_range", int(start[k]), int(end[k])))
                    if nr_r[k] >= 0:
                        nacked.append((int(start[k]) % L, jvm.call("roundLeader", C.c_int32, 2, int(rnd[k])), wire.mencius_encode("leader_nack", int(nr_r[k]))))
            i = j
        return sent, nacked

    cmd = lambda s, tag: bytes.fromhex("0a") + bytes([len(b"%s %d" % (tag, s))]) + b"%s %d" % (tag, s)
    p2a = lambda s, r, tag: wire.mencius_encode("proxy_leader_phase2a", s, r, cmd(s, tag))
    rng = lambda a, b, r: wire.mencius_encode("proxy_leader_phase2a_noop_range", a, b, r)
    # 1: Phase 1 in round 0, everywhere
    for lg in range(L):
        for ag in range(A):
            for a in range(R):
                reply = phase1a_at(lg * A + ag, a, wire.mencius_encode("acceptor_phase1a", 0, 0))
                assert reply.hex() == "0a06" + "08%02x" % ag + "10%02x" % a + "1800"
    # 2: group 0 proposes in slots 0, 2, .. 38; group 1 skips slots 1 .. 19 (a range), then proposes 21 .. 39
    msgs = ([p2a(s, 0, b"set") for s in range(0, 40, 2)][:10] + [rng(1, 21, 0)] + [p2a(s, 0, b"set") for s in range(20, 40, 2)] +
            [p2a(s, 0, b"set") for s in range(21, 41, 2)])
    sent, nacked = burst(msgs)
    assert len(sent) == 10 + 1 + 10 + 10 and not nacked
    d = wire.mencius_decode_replica_inbound(sent)
    assert d["kind"].tolist().count(wire.CHOSEN_NOOP_RANGE) == 1 and d["slot"][10] == 1 and d["slot_end"][10] == 21
    # 3: leader 1 of group 1 starts Phase 1 in round 1 with two acceptors of each of the group's acceptor groups
    for ag in range(A):
        for a in (0, 1):
            d = wire.decode_leader_inbound([phase1a_at(1 * A + ag, a, wire.mencius_encode("acceptor_phase1a", 1, 9))])
            assert d["kind"].tolist() == [wire.PHASE1B] and d["round"].tolist() == [1] and d["group_index"].tolist() == [ag]
            slots = d["info_slot"].tolist()
            assert slots == [s for s in range(9, 41, 2) if group_of(s) == A + ag] and set(d["info_vote_round"].tolist()) == {0}
    # 4: the old leader of group 1 (round 0) and group 0 (undisturbed), dense delivery
    sent, nacked = burst([p2a(41, 0, b"stale"), rng(43, 61, 0), p2a(40, 0, b"set"), p2a(42, 0, b"set")])
    assert [wire.mencius_decode_replica_inbound([x])["slot"][0] for x in sent] == [40, 42]
    assert nacked == [(1, 0, bytes.fromhex("3a020801"))] * 2                     # LeaderInbound.nack = field 7, Nack(round 1)
    # 5: the new leader finishes Phase 1 with everybody, re-proposes what the acceptors reported beyond the watermark, and
    #    skips on in its round
    for ag in range(A):
        phase1a_at(1 * A + ag, 2, wire.mencius_encode("acceptor_phase1a", 1, 9))
    sent, nacked = burst([p2a(41, 1, b"stale"), rng(43, 61, 1), p2a(61, 1, b"set")])   # (slot 41: the value acceptor 2 voted for)
    assert len(sent) == 3 and not nacked
    d = wire.mencius_decode_replica_inbound(sent)
    assert d["slot"].tolist() == [41, 43, 61] and d["slot_end"].tolist() == [-1, 61, -1]
    np.testing.assert_array_equal(np.asarray(ref.state_digest()), np.asarray(_digest(jvm, h)))
    assert jvm.call("destroy", C.c_int32, h) == 0


def _digest(jvm, h):
    import frankenpaxos_amd as fa
    out = (C.c_uint64 * 8)()
    assert fa.lib().fpx_state_digest(C.c_void_p(h), out) == 0
    return list(out)


@pytest.mark.gpu
def test_an_epaxos_slow_path_commit_on_wire_bytes_through_the_shim(jvm, oracle):
    """Two conflicting commands led by different replicas meet at one replica in the other order: the PreAcceptOks of
    instance A differ, its leader takes the slow path (epaxos/Replica.scala:1291-1419, 796-813) -- Accept, AcceptOk, Commit.
    Every message between the replica addresses crosses as ReplicaInbound BYTES: encoded by wireEpaxosEncodeReplicaInbound,
    decoded by wireEpaxosDecodeReplicaInbound, handled by the natives GpuEPaxosReplica batches a tick onto
    (epxHandlePreaccept = Replica.handlePreAccept :1159-1289, epxAccept = transitionToAcceptPhase + handleAccept +
    handleAcceptOk :732-792, 1421-1565).  n = 5 (f = 2), all five replicas hosted by one libfpx context.  Every integer
    equals the oracle fed the same calls, every byte string the python codec (pinned on google.protobuf)."""
    from frankenpaxos_amd import wire

    n, keys, NI = 5, 4, 16
    h = jvm.call("epxCreateWithLog", C.c_int64, n, keys, 0, NI)
    assert h > 0
    ref = oracle.EPaxos(n, keys, num_instances=NI)
    i32 = lambda a: jvm.arr(np.asarray(a, np.int32))
    i8 = lambda a: jvm.arr(np.asarray(a, np.int8))
    command = {7: bytes.fromhex("0a0b") + b"set k1 = 07", 9: bytes.fromhex("0a0b") + b"set k1 = 09"}   # triple id -> CommandOrNoop

    def enc(kind, instance, ballot=(-1, -1), replica_index=-1, seq=-1, triple=None, deps=None, values=()):
        """through the native; == the python codec"""
        head = i32([kind, instance[0], instance[1], ballot[0], ballot[1], replica_index, seq, -1, -1, -1,
                    -1 if triple is None else 0])
        cmd = command[triple] if triple is not None else b""
        out = _direct(jvm, b"\0" * 512)
        vals = i32([v[0] for v in values] + [v[1] for v in values]) if values else None
        ln = jvm.call("wireEpaxosEncodeReplicaInbound", C.c_int64, out, head, _direct(jvm, cmd), 0, len(cmd),
                      -1 if deps is None else n, None if deps is None else i32(deps), len(values), vals)
        assert ln > 0
        got = _dbytes(jvm, out, ln)
        assert got == wire.epaxos_encode(kind, instance, ballot, replica_index, None if seq < 0 else seq,
                                         command=False if triple is None else cmd, deps=deps, values=values)
        return got

    def dec(msgs):
        """through the native; == the python codec"""
        buf, off = wire.pack(msgs)
        k = len(msgs)
        fields, co, dw = jvm.arr(np.zeros(13 * k, np.int32)), jvm.arr(np.zeros(k, np.int64)), jvm.arr(np.zeros(k * n, np.int32))
        vo, vals, bad = jvm.arr(np.zeros(k + 1, np.int64)), jvm.arr(np.zeros(2 * 16, np.int32)), jvm.arr(np.zeros(1, np.int32))
        assert jvm.call("wireEpaxosDecodeReplicaInbound", C.c_int32, _direct(jvm, buf), jvm.arr(off), k, n, fields, co, dw, vo, 16, vals, bad) == 0
        f = jvm.read(fields, np.int32, 13 * k).reshape(13, k)
        want = wire.epaxos_decode_replica_inbound(msgs, max_replicas=n)
        for j, name in enumerate(("kind", "instance_leader", "instance_number", "ballot_ordering", "ballot_replica", "replica_index",
                                  "sequence_number", "vote_ballot_ordering", "vote_ballot_replica", "status", "is_noop", "cmd_len",
                                  "deps_num_replicas")):
            np.testing.assert_array_equal(f[j], want[name], err_msg=name)
        deps = jvm.read(dw, np.int32, k * n).reshape(k, n)
        np.testing.assert_array_equal(deps, want["deps_watermark"])
        return f, deps, jvm.read(co, np.int64, k), buf

    def preaccept_at(msg, targets, key, triple):
        """a PreAccept's bytes delivered to the replicas `targets` -> {replica: PreAcceptOk bytes}"""
        f, deps, co, buf = dec([msg])
        assert f[0][0] == wire.EPX_PRE_ACCEPT and bytes(buf[co[0]:co[0] + f[11][0]]) == command[triple]
        L, x, bo, br = (int(f[j][0]) for j in (1, 2, 3, 4))
        mask = sum(1 << r for r in targets)
        rep, nb = jvm.arr(np.zeros(4, np.int8)), jvm.arr(np.zeros(1, np.int32))
        rd, ret = jvm.arr(np.zeros(n * n, np.int32)), jvm.arr(np.zeros(2 * n, np.int32))
        assert jvm.call("epxHandlePreaccept", C.c_int32, h, 1, n, i32([L]), i32([x]), i32([bo]), i32([br]), i32([key]), i8([1]),
                        i32([triple]), i32(deps[0]), i32([0]), i8([mask]), rep, nb, rd, ret) == 0
        want = ref.handle_preaccept([L], [x], [bo], [br], [key], [1], [triple], deps[:1], [0], [mask])
        assert want[0] == 0
        got = jvm.read(rep, np.int8, 4).view(np.uint8).tolist()
        assert got == [int(want[k][0]) for k in (1, 2, 3, 4)] and got[0] == mask        # every target processed it
        reply_deps = jvm.read(rd, np.int32, n * n).reshape(n, n)
        np.testing.assert_array_equal(reply_deps, want[6][0])
        ends = jvm.read(ret, np.int32, 2 * n)[:n]
        np.testing.assert_array_equal(ends, want[7][0])
        assert not ends.any()
        return {r: enc(wire.EPX_PRE_ACCEPT_OK, (L, x), (bo, br), replica_index=r, seq=0, deps=reply_deps[r].tolist()) for r in targets}

    A, B = (0, 0), (4, 0)
    none = [0] * n
    # the two leaders pre-accept their own instances (transitionToPreAcceptPhase: conflicts in THEIR index, none yet)
    own_a = preaccept_at(enc(wire.EPX_PRE_ACCEPT, A, (0, 0), seq=0, triple=7, deps=none), [0], 1, 7)
    own_b = preaccept_at(enc(wire.EPX_PRE_ACCEPT, B, (0, 4), seq=0, triple=9, deps=none), [4], 1, 9)
    d_a = dec([own_a[0]])[1][0].tolist()
    d_b = dec([own_b[4]])[1][0].tolist()
    assert d_a == none and d_b == none
    # B's PreAccept reaches replica 1 first ...
    ok_b = preaccept_at(enc(wire.EPX_PRE_ACCEPT, B, (0, 4), seq=0, triple=9, deps=d_b), [1], 1, 9)
    assert dec([ok_b[1]])[1][0].tolist() == none
    # ... then A's reaches replicas 1, 2, 3: replica 1 already holds B, which conflicts (two sets of one key)
    ok_a = preaccept_at(enc(wire.EPX_PRE_ACCEPT, A, (0, 0), seq=0, triple=7, deps=d_a), [1, 2, 3], 1, 7)
    f, answers, _, _ = dec([ok_a[r] for r in (1, 2, 3)])
    assert f[0].tolist() == [wire.EPX_PRE_ACCEPT_OK] * 3 and f[5].tolist() == [1, 2, 3]
    assert answers[0].tolist() == [0, 0, 0, 0, 1] and answers[1].tolist() == none and answers[2].tolist() == none
    # Replica.handlePreAcceptOk (:1376-1419): n - 2 = 3 answers, not all equal -> preAcceptingSlowPath: the union, Accept
    union = np.maximum.reduce([answers[0], answers[1], answers[2], np.asarray(d_a)]).tolist()
    assert union == [0, 0, 0, 0, 1]
    accept = enc(wire.EPX_ACCEPT, A, (0, 0), seq=0, triple=7, deps=union)
    f, deps, co, buf = dec([accept])
    assert f[0][0] == wire.EPX_ACCEPT and deps[0].tolist() == union
    rep4, nb = jvm.arr(np.zeros(4, np.int8)), jvm.arr(np.zeros(1, np.int32))
    # to f = 2 other replicas: with the proposer's own vote that is the slow quorum f + 1
    assert jvm.call("epxAccept", C.c_int32, h, 1, i32([0]), i32([0]), i32([0]), i32([0]), i32([7]), i32([1]), i8([1]),
                    i8([0b00110]), rep4, nb) == 0
    want = ref.accept([0], [0], [0], [0], [7], [0b00110], [1], [1])
    got = jvm.read(rep4, np.int8, 4).view(np.uint8).tolist()
    assert want[0] == 0 and got == [int(want[k][0]) for k in (1, 2, 3, 5)]
    assert got[0] == 0b00111 and got[3] == 1                  # the proposer's own vote + two AcceptOks = f + 1: committed
    oks = [enc(wire.EPX_ACCEPT_OK, A, (0, 0), replica_index=r) for r in (1, 2)]
    f, _, _, _ = dec(oks)
    assert f[0].tolist() == [wire.EPX_ACCEPT_OK] * 2 and f[5].tolist() == [1, 2]
    commit = enc(wire.EPX_COMMIT, A, seq=0, triple=7, deps=union)
    f, deps, _, _ = dec([commit])
    assert f[0][0] == wire.EPX_COMMIT and deps[0].tolist() == union
    # the command logs and the conflict indices of all five replicas == the oracle's
    entry = jvm.arr(np.zeros(6 + n, np.int32))
    for inst in (A, B):
        for r in range(n):
            assert jvm.call("epxReadCmdlog", C.c_int32, h, n, r, inst[0], inst[1], entry) == 0
            got = jvm.read(entry, np.int32, 6 + n)
            assert tuple(got[:5]) == ref.read_cmdlog(r, *inst)
            dd, end = ref.read_cmdlog_deps(r, *inst)
            assert got[5:5 + n].tolist() == dd.tolist() and got[5 + n] == end
    assert ref.read_cmdlog(3, *A)[0] == 4 and ref.read_cmdlog(0, *A)[0] == 4         # CommittedEntry everywhere
    import frankenpaxos_amd as fa
    for r in range(n):
        gg, ss = (C.c_int32 * n)(), (C.c_int32 * n)()
        assert fa.lib().fpx_epx_read_index(C.c_void_p(h), r, 1, gg, ss) == 0
        g2, s2 = ref.read_index(r, 1)
        assert list(gg) == g2.tolist() and list(ss) == s2.tolist()
    assert jvm.call("epxDestroy", C.c_int32, h) == 0
