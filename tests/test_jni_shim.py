"""The JNI shim cannot be built for real here (no JDK): type-check it against include/fpx.h with a
stand-in <jni.h> (tests/jni_stub) so that a signature drift between the shim and the C ABI is caught,
and check that every native method the Scala side declares has its Java_... function in the shim."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI = os.path.join(ROOT, "frankenpaxos_amd", "jni")


def test_shim_type_checks_against_the_c_abi(tmp_path):
    cmd = ["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-parameter", "-Wno-comment",
           "-I" + os.path.join(ROOT, "tests", "jni_stub"), os.path.join(JNI, "fpx_jni.c")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_scala_natives_have_shim_functions():
    scala = open(os.path.join(JNI, "Native.scala")).read()
    shim = open(os.path.join(JNI, "fpx_jni.c")).read()
    natives = re.findall(r"@native def (\w+)\(", scala)
    assert len(natives) >= 7
    for name in natives:
        assert "Java_frankenpaxos_gpu_Native_" + name in shim, name


def test_every_pinned_array_is_released():
    """GetPrimitiveArrayCritical without its Release would wedge the JVM's garbage collector"""
    shim = open(os.path.join(JNI, "fpx_jni.c")).read()
    bodies = re.split(r"\nJNIEXPORT ", shim)[1:]
    assert len(bodies) >= 19
    for body in bodies:
        name = re.search(r"Java_frankenpaxos_gpu_Native_(\w+)", body).group(1)
        released = re.findall(r"UNPIN\(env, (\w+),", body)
        only_pins = [a for a in re.findall(r"(?<!UN)PIN\(env, (\w+)\)", body)]
        assert sorted(only_pins) == sorted(released), (name, only_pins, released)
