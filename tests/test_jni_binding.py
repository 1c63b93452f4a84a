"""The Java binding, the JNI shim and the C header must agree -- checked at the text level, because the build image has
no JDK: every `native` method of NativeNormalEquationEngine has exactly one Java_... function with matching parameter
types, every structure field the shim looks up exists in ProblemDescription with that type, every jaicov_neq_* function
the shim calls is declared in include/jaicov_neq.h (and exported by the library when it has been built), and the shim
compiles (syntax and types) against a minimal jni.h."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "java", "org", "applied_geodesy", "adjustment", "bundle", "nativeengine", "NativeNormalEquationEngine.java")
SHIM = os.path.join(ROOT, "java", "jni", "jaicov_jni.c")
HEADER = os.path.join(ROOT, "include", "jaicov_neq.h")

JTYPE = {"long": "jlong", "int": "jint", "double": "jdouble", "boolean": "jboolean", "double[]": "jdoubleArray",
         "int[]": "jintArray", "long[]": "jlongArray", "String": "jstring", "void": "void", "ProblemDescription": "jobject"}


def java_natives():
    src = open(JAVA).read()
    out = {}
    for ret, name, args in re.findall(r"private static native ([\w\[\]]+) (\w+)\((.*?)\);", src):
        params = [a.strip().rsplit(" ", 1)[0] for a in args.split(",")] if args.strip() else []
        assert name not in out, f"overloaded native {name}: JNI short names would clash"
        out[name] = (ret, params)
    return out


def shim_functions():
    src = open(SHIM).read()
    out = {}
    for ret, name, args in re.findall(r"JNIEXPORT (\w+) JNICALL NAT\((\w+)\)\((.*?)\)\s*\{", src, flags=re.S):
        params = [a.strip().rsplit(" ", 1)[0].strip() for a in args.split(",")]
        assert params[0] == "JNIEnv" or params[0].startswith("JNIEnv"), (name, params)
        assert params[1] == "jclass", (name, params)          # static natives
        assert name not in out
        out[name] = (ret, params[2:])
    return out


def test_natives_and_shim_match_one_to_one():
    jn, sh = java_natives(), shim_functions()
    assert len(jn) >= 20
    assert set(jn) == set(sh), (sorted(set(jn) - set(sh)), sorted(set(sh) - set(jn)))
    for name, (ret, params) in jn.items():
        cret, cparams = sh[name]
        assert JTYPE[ret] == cret, (name, ret, cret)
        assert [JTYPE[p] for p in params] == cparams, (name, params, cparams)
    src = open(SHIM).read()
    assert "org_applied_1geodesy_adjustment_bundle_nativeengine_NativeNormalEquationEngine_" in src   # package and class of JAVA
    assert "package org.applied_geodesy.adjustment.bundle.nativeengine;" in open(JAVA).read()


def test_shim_field_lookups_exist_in_problem_description():
    java = open(JAVA).read()
    body = java[java.index("class ProblemDescription"):]
    body = body[:body.index("}")]
    fields = {}
    for typ, names in re.findall(r"public ([\w\[\]]+) ([\w, ]+);", body):
        for n in names.split(","):
            fields[n.strip()] = typ
    sig = {"I": "int[]", "J": "long[]", "B": "byte[]", "D": "double[]"}
    shim = open(SHIM).read()
    looked = re.findall(r'\{"(\w+)", \'([IJBD])\'\}', shim)
    assert len(looked) == 30
    for name, kind in looked:
        assert fields.get(name) == sig[kind], (name, kind, fields.get(name))
    for name in re.findall(r'GetFieldID\(e, c, "(\w+)", "I"\)', shim):
        assert fields.get(name) == "int", name
    # every pointer member of jaicov_problem_desc is filled
    hdr = open(HEADER).read()
    desc = hdr[hdr.index("typedef struct jaicov_problem_desc {"):hdr.index("} jaicov_problem_desc;")]
    members = [m for grp in re.findall(r"const \w+\s*\*([\w, *]+);", desc) for m in re.split(r"[,\s*]+", grp) if m]
    assert len(members) == 30
    for m in members:
        assert re.search(r"p\.%s = " % m, shim), f"jaicov_problem_desc.{m} is not set by the shim"


def test_every_c_abi_call_of_the_shim_is_declared_and_exported():
    hdr = open(HEADER).read()
    declared = set(re.findall(r"\b(jaicov_neq_\w+)\s*\(", hdr))
    used = set(re.findall(r"\b(jaicov_neq_\w+)\s*\(", open(SHIM).read()))
    assert used and used <= declared, sorted(used - declared)
    # what VERDICT r1 asked the shim to reach
    for f in ("jaicov_neq_accumulate2", "jaicov_neq_finalize", "jaicov_neq_reduce_buffer", "jaicov_neq_get_normal",
              "jaicov_neq_get_cofactor_sub", "jaicov_neq_estimate", "jaicov_neq_cancel"):
        assert f in used, f
    lib = os.path.join(ROOT, "bundle-adjustment_amd", "csrc", "libjaicov_neq.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
        for f in used:
            assert re.search(r"\bT %s\b" % f, syms), f"{f} is not exported by libjaicov_neq.so"


def test_shim_compiles_against_a_minimal_jni_header():
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "jni_stub"),
                        "-I", os.path.join(ROOT, "include"), SHIM], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the two bugs of the first version must not come back: a critical pointer released with NULL, a dropped create status
    src = open(SHIM).read()
    assert not re.search(r"ReleasePrimitiveArrayCritical\(e, [^,]+, NULL", src)
    assert "rc != JAICOV_OK" in src and "throw_status(e, rc, msg)" in src
