"""Generates tests/golden/config_parse.json with the REFERENCE's own parser (oracle/_ref/libreadconfig_ref.so =
/root/reference/src/tools/readconfig.c compiled by oracle/Makefile) over the texts in tests/config_cases.py.
Run from the repo root:  python tests/golden/make_config_golden.py"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))
import config_cases as cc      # noqa: E402
import ref_readconfig as rr    # noqa: E402


def write_cases(d):
    for name, text in cc.CASES.items():
        with open(os.path.join(d, name + ".cfg"), "w", newline="") as f:
            f.write(text)
    for rel, text in cc.INCLUDE_FILES.items():
        os.makedirs(os.path.dirname(os.path.join(d, "inc", rel)), exist_ok=True)
        with open(os.path.join(d, "inc", rel), "w", newline="") as f:
            f.write(text)
    for name, text in cc.ERROR_CASES.items():
        with open(os.path.join(d, "err_" + name + ".cfg"), "w", newline="") as f:
            f.write(text)


def scrub(obj, d):
    """The temporary directory must not leak into the fixture."""
    s = json.dumps(obj)
    return json.loads(s.replace(d, "@dir"))


def main():
    assert rr.lib() is not None, "build oracle/_ref first (make -C oracle)"
    out = {"files": {}, "argv": {}, "include": None, "errors": {}}
    with tempfile.TemporaryDirectory() as d:
        write_cases(d)
        for name in cc.CASES:
            out["files"][name] = scrub(rr.dump(cfg_path=os.path.join(d, name + ".cfg")), d)
        for name, argv in cc.ARGV_CASES.items():
            out["argv"][name] = scrub(rr.dump(argv=[a.replace("@dir", d) for a in argv]), d)
        out["include"] = scrub(rr.dump(cfg_path=os.path.join(d, "inc", "main.cfg")), d)
        for name in cc.ERROR_CASES:
            out["errors"][name] = rr.dump(cfg_path=os.path.join(d, "err_" + name + ".cfg"))
    with open(os.path.join(HERE, "config_parse.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote config_parse.json:", {k: (len(v) if v else 0) for k, v in out.items()})


if __name__ == "__main__":
    main()
