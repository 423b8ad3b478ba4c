"""The declaration stubs of tests/shim_stubs/ transcribe the reference interfaces the shim subclasses; this test keeps them from
drifting: every `virtual` declaration of a stub header, every CREATE_CHRONOMETER member and every inherited `_member` the shim
touches is looked up in the TEXT of the reference's own headers (study of the text, nothing is compiled or imported).  Runs in the
build container only — /root/reference does not travel to the GPU box."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
STUBS = os.path.join(ROOT, "tests", "shim_stubs")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")

# stub header -> reference headers that declare the same classes (base classes and the macros they are made with included)
PAIRS = {
    "framepoint_generation/stereo_framepoint_generator.h": ["framepoint_generation/base_framepoint_generator.h",
                                                            "framepoint_generation/stereo_framepoint_generator.h", "types/definitions.h"],
    "aligners/stereouv_aligner.h": ["aligners/base_aligner.h", "aligners/base_frame_aligner.h", "aligners/stereouv_aligner.h",
                                    "types/definitions.h"],
}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def canonical_param(p):
    p = p.split("=")[0].strip()                       # default values are not part of the signature
    tokens = re.findall(r"[A-Za-z_][A-Za-z_0-9:<>]*|[*&]", p)
    if len(tokens) > 1 and re.match(r"^(_\w+|\w+_)$", tokens[-1]):   # the reference names every parameter name_ (or _name)
        tokens = tokens[:-1]
    return " ".join(tokens).replace(" *", "*").replace(" &", "&")


def virtual_signatures(text):
    """{(return type, name, (parameter types), const)} of every `virtual` member function declared in `text`."""
    out = set()
    text = strip_comments(text).replace("\\\n", "\n")   # a macro body line by line
    for m in re.finditer(r"\bvirtual\b", text):
        start = text.rfind("\n", 0, m.start()) + 1
        head = text[start:m.start()].strip()           # `void virtual track(` puts the return type in front
        rest = text[m.end():]
        par = rest.find("(")
        if par < 0:
            continue
        decl = (head + " " + rest[:par]).split()
        if not decl or decl[-1].startswith("~"):       # destructors carry the class name
            continue
        depth, i = 0, par
        while i < len(rest):
            depth += rest[i] == "("
            depth -= rest[i] == ")"
            if depth == 0:
                break
            i += 1
        params = rest[par + 1:i].strip()
        tail = rest[i + 1:i + 40]
        const = bool(re.match(r"\s*const\b", tail))
        plist = tuple(canonical_param(p) for p in params.split(",")) if params else ()
        ret = " ".join(t for t in decl[:-1] if t not in ("inline", "public:", "protected:", "private:"))
        out.add((ret, decl[-1], plist, const))
    return out


def read(path):
    with open(path) as f:
        return f.read()


@pytest.mark.parametrize("stub", sorted(PAIRS))
def test_stub_virtuals_match_the_reference_headers(stub):
    ref_text = "\n".join(read(os.path.join(REF, h)) for h in PAIRS[stub])
    ref_sigs = virtual_signatures(ref_text)
    # PROSLAM_MAKE_PROCESSING_CLASS declares `virtual void configure();` for every processing class (definitions.h:31-37)
    stub_sigs = virtual_signatures(read(os.path.join(STUBS, stub)))
    assert len(stub_sigs) >= 4, stub_sigs
    missing = sorted(s for s in stub_sigs if s not in ref_sigs)
    assert not missing, "stub declares virtuals the reference headers do not: %s\nreference has: %s" % (missing, sorted(ref_sigs))


def test_chronometer_members_exist_where_the_stubs_put_them():
    ref_macro = re.sub(r"\s+", " ", re.search(r"#define CREATE_CHRONOMETER\(NAME\)(.*?)\n\s*#define", read(os.path.join(REF, "types/definitions.h")), re.S).group(1).replace("\\", " ")).strip()
    stub_macro = re.sub(r"\s+", " ", re.search(r"#define CREATE_CHRONOMETER\(NAME\)(.*?)\n\n", read(os.path.join(STUBS, "types/definitions.h")), re.S).group(1).replace("\\", " ")).strip()
    assert ref_macro == stub_macro
    for header, names in {"framepoint_generation/base_framepoint_generator.h": ("keypoint_detection", "descriptor_extraction"),
                          "framepoint_generation/stereo_framepoint_generator.h": ("point_triangulation",)}.items():
        text = strip_comments(read(os.path.join(REF, header)))
        for n in names:
            assert "CREATE_CHRONOMETER(%s)" % n in text, (header, n)
    stub = read(os.path.join(STUBS, "framepoint_generation/stereo_framepoint_generator.h"))
    for n in ("keypoint_detection", "descriptor_extraction", "point_triangulation"):
        assert "CREATE_CHRONOMETER(%s)" % n in stub
    # slam_assembly.cpp:709-719 is the reader these members exist for
    report = read("/root/reference/src/system/slam_assembly.cpp")
    for n in ("keypoint_detection", "descriptor_extraction", "point_triangulation"):
        assert "getTimeConsumptionSeconds_%s" % n in report


def test_inherited_members_the_shim_touches_exist_in_the_reference():
    shim = strip_comments(read(os.path.join(ROOT, "shim", "proslam_hip_plugin.h")))
    own = {"_hip", "_pruned", "_computed", "_descriptors_pending", "_rec_kp", "_rec_meta", "_rec_cam", "_rec_desc", "_timers_enabled", "_features_left", "_features_right", "_pixel_left", "_pixel_right", "_last_info"}
    used = set(re.findall(r"(?<![\w.>])(_[a-z][a-z_0-9]*)\b", shim)) - own
    ref_text = "\n".join(strip_comments(read(os.path.join(REF, h))) for hs in PAIRS.values() for h in hs)
    expanded = ref_text + " " + " ".join("_time_consumption_seconds_" + n for n in re.findall(r"CREATE_CHRONOMETER\((\w+)\)", ref_text))
    missing = sorted(m for m in used if not re.search(r"\b%s\b" % re.escape(m), expanded))
    assert len(used) > 15, used
    assert not missing, "the shim uses members no reference header declares: %s" % missing
