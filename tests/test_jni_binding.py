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
         "int[]": "jintArray", "long[]": "jlongArray", "String": "jstring", "void": "void", "ProblemDescription": "jobject", "EngineOptions": "jobject"}


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


# C-ABI functions the Java binding deliberately does not reach, each with its reason.  Everything else the header declares must be called by the shim.
NOT_BOUND = {
    "jaicov_neq_accumulate": "superseded by jaicov_neq_accumulate2 (the same call without the damping value; kept for round-1 clients)",
}
# members of jaicov_engine_options that have no Java field, each with its reason
OPTION_NOT_BOUND = {"struct_size": "filled by the shim (sizeof)", "reserved": "reserved words, must be zero"}


def test_every_export_of_the_header_is_bound_or_excluded_with_a_reason():
    """the Java half of the boundary lagged the header (options and three exchange buffers added since round 2)."""
    hdr = open(HEADER).read()
    declared = set(re.findall(r"^(?:int|void|size_t|const char \*)\s*(jaicov_neq_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) == 34, sorted(declared)      # the list in include/jaicov_neq.h; a new export must be bound or excluded here
    used = set(re.findall(r"\b(jaicov_neq_\w+)\s*\(", open(SHIM).read()))
    missing = declared - used - set(NOT_BOUND)
    assert not missing, f"declared in the header, neither bound by the shim nor excluded: {sorted(missing)}"
    assert not (set(NOT_BOUND) & used), "an excluded function is bound after all: drop it from NOT_BOUND"
    for f in ("jaicov_neq_eo_step_buffer", "jaicov_neq_expansion_buffer", "jaicov_neq_reduce_buffer_async"):
        assert f in used, f
    # the library exports every declared function
    lib = os.path.join(ROOT, "bundle-adjustment_amd", "csrc", "libjaicov_neq.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
        for f in declared:
            assert re.search(r"\bT %s\b" % f, syms), f"{f} is declared but not exported"


def test_every_engine_option_is_settable_from_java():
    hdr = open(HEADER).read()
    body = hdr[hdr.index("typedef struct jaicov_engine_options {"):hdr.index("} jaicov_engine_options;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    members = [m for grp in re.findall(r"^\s*u?int32_t\s+([\w, \[\]]+);", body, flags=re.M) for m in re.split(r"[,\s]+", grp) if m]
    members = [re.sub(r"\[.*", "", m) for m in members]
    assert set(OPTION_NOT_BOUND) <= set(members) and len(members) >= 14, members
    shim = open(SHIM).read()
    java = open(JAVA).read()
    opt = java[java.index("class EngineOptions"):]
    opt = opt[:opt.index("public static EngineOptions fromSystemProperties")]
    jfields = {}
    for typ, names in re.findall(r"public (int|boolean) ([\w, =\-0-9truefals]+);", opt):
        for n in names.split(","):
            jfields[n.split("=")[0].strip()] = typ
    table = dict(re.findall(r'\{"(\w+)", offsetof\(jaicov_engine_options, (\w+)\)\}', shim))     # Java int field -> C member
    bound = set(table.values()) | {"apply_shared"}
    for m in members:
        assert m in bound or m in OPTION_NOT_BOUND, f"jaicov_engine_options.{m} cannot be set from Java"
    for jname, cname in table.items():
        assert jfields.get(jname) == "int", (jname, jfields.get(jname))
        assert jname.lower() == cname.replace("_", ""), (jname, cname)      # camelCase of the C name
    assert jfields.get("applyShared") == "boolean" and 'GetFieldID(e, oc, "applyShared", "Z")' in shim
    assert set(jfields) == set(table) | {"applyShared"}, sorted(set(jfields) ^ (set(table) | {"applyShared"}))
    # the six options added since round 2, by name (What's missing 2)
    for m in ("deterministic", "refinement", "ordinary_group_elimination", "dispersion_refinement", "reduced_reference_quirk", "expansion_exchange"):
        assert m in bound, m
    # the patched BundleAdjustment passes the reference-visible ones from system properties, next to the switch itself
    ba = open(os.path.join(ROOT, "java", "patch", "BundleAdjustment.native.patch")).read()
    assert "EngineOptions.fromSystemProperties()" in ba
    for prop in ("native.reducedReferenceQuirk", "native.deterministic"):
        assert prop in opt + java[java.index("fromSystemProperties"):], prop


def test_documents_state_the_number_of_natives_that_exist():
    n = len(re.findall(r"^JNIEXPORT", open(SHIM).read(), flags=re.M))
    assert n == len(java_natives())
    for doc in ("README.md", "INTEGRATION.md", "DESIGN.md"):
        for claimed in re.findall(r"(\d+) natives", open(os.path.join(ROOT, doc)).read()):
            assert int(claimed) == n, (doc, claimed, n)


def test_shim_compiles_against_a_minimal_jni_header():
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "jni_stub"),
                        "-I", os.path.join(ROOT, "include"), SHIM], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the two bugs of the first version must not come back: a critical pointer released with NULL, a dropped create status
    src = open(SHIM).read()
    assert not re.search(r"ReleasePrimitiveArrayCritical\(e, [^,]+, NULL", src)
    assert "rc != JAICOV_OK" in src and "throw_status(e, rc, msg)" in src


# ---- the Java side of the flattening (java/.../ProblemFlattener.java + java/patch/*.patch), checked as text ---------------------
FLAT = os.path.join(ROOT, "java", "org", "applied_geodesy", "adjustment", "bundle", "nativeengine", "ProblemFlattener.java")
PATCH = os.path.join(ROOT, "java", "patch")


def _description_fields():
    java = open(JAVA).read()
    body = java[java.index("class ProblemDescription"):]
    body = body[:body.index("}")]
    fields = {}
    for typ, names in re.findall(r"public ([\w\[\]]+) ([\w, ]+);", body):
        for n in names.split(","):
            fields[n.strip()] = typ
    return fields


def test_flattener_fills_every_field_of_the_problem_description():
    src = open(FLAT).read()
    fields = _description_fields()
    assigned = set(re.findall(r"\bd\.(\w+)\s*=[^=]", src))
    assert assigned == set(fields), (sorted(set(fields) - assigned), sorted(assigned - set(fields)))
    # element types: arrays are created / converted with the type the description declares
    conv = {"int[]": ("new int[", "toIntArray("), "long[]": ("toLongArray(",), "double[]": ("new double[", "toDoubleArray(", "concat("),
            "byte[]": ("new byte[",)}
    for name, typ in fields.items():
        if typ == "int":
            continue
        rhs = re.search(r"\bd\.%s\s*=\s*([^;]+);" % name, src).group(1)
        assert any(rhs.strip().startswith(c) for c in conv[typ]), (name, typ, rhs)
    # UnknownParameter.java:27: fixed = Integer.MAX_VALUE -> JAICOV_COL_FIXED (-1); unset (-1) never reaches the engine as a column
    assert re.search(r"c == Integer\.MAX_VALUE \|\| c < 0\) \? -1 : c", src)
    # the walk follows prepareUnknownParameters(): points, interior orientation, distortion, exterior orientation, image points,
    # scale bars, directly observed groups -- the slot order of include/jaicov_neq.h
    order = [src.index(k) for k in ("d.pointColumn = new", "d.interiorColumn = new", "d.distortionKind =", "d.exteriorColumn =",
                                    "d.imagePointImage =", "d.scaleBarA = new", "d.directSlot =")]
    assert order == sorted(order)
    # datum bits and distortion kinds equal the header's enums
    hdr = open(HEADER).read()
    for name, val in re.findall(r"JAICOV_DATUM_(\w+) = (\d+)", hdr):
        assert re.search(r"DATUM_%s = %s\b" % (name, val), src), name
    kinds = dict(re.findall(r"JAICOV_DIST_(\w+)\s*= (\d+)", hdr))
    for name, val in kinds.items():
        assert re.search(r"\b%s = %s\b" % (name, val), src), name


def test_flattener_and_patches_agree_with_the_engine_class_and_with_each_other():
    src = open(FLAT).read()
    eng = open(JAVA).read()
    ba = open(os.path.join(PATCH, "BundleAdjustment.native.patch")).read()
    img = open(os.path.join(PATCH, "Image.dispersion.patch")).read()
    dop = open(os.path.join(PATCH, "DirectlyObservedParameterGroup.dispersion.patch")).read()
    # API additions used by the flattener are the ones the patches add
    assert "image.getDispersion()" in src and "public UpperSPDPackMatrix getDispersion()" in img
    assert "public void setDispersion(UpperSPDPackMatrix dispersionMatrix)" in img
    assert "group.getDispersionMatrix()" in src and "public UpperSPDPackMatrix getDispersionMatrix()" in dop
    # every engine method the BundleAdjustment patch calls exists in the binding class with that name
    public = set(re.findall(r"public (?:static )?[\w\[\]<>?]+ (\w+)\(", eng))
    for m in set(re.findall(r"this\.engine\.(\w+)\(", ba)):
        assert m in public, m
    for m in set(re.findall(r"this\.flattener\.(\w+)\(", ba)):
        assert re.search(r"public [\w\[\].]+ %s\(" % m, src), m
    # the flatten() call passes what ProblemFlattener.flatten() takes, in that order
    sig = re.search(r"ProblemDescription flatten\((.*?)\)\s*\{", src, flags=re.S).group(1)
    names = [a.strip().rsplit(" ", 1)[1] for a in sig.replace("\n", " ").split(",")]
    call = re.search(r"this\.flattener\.flatten\((.*?)\);", ba, flags=re.S).group(1)
    passed = [a.strip().replace("this.", "") for a in call.replace("\n", " ").replace("+", "").split(",")]
    assert names == passed, (names, passed)
    # MatrixInversion.FULL is run as JAICOV_INVERT_FULL_EXPANDED (3), like jaicov_neq_estimate and the C++ host do
    hdr = open(HEADER).read()
    assert "#define JAICOV_INVERT_FULL_EXPANDED 3" in hdr and "MatrixInversion.FULL ? 3" in ba
