"""CPU: structural guarantees of the product tree -- it never touches the oracle or the reference, and it has no
CPU compute fallback."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT_DIRS = ["plonk_gadgets_amd", "include"]


def product_files():
    for d in PRODUCT_DIRS:
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".hpp", ".hip", ".h", ".inc")):
                    yield os.path.join(base, f)


def test_product_never_imports_oracle_or_reads_reference():
    bad = []
    for path in product_files():
        text = open(path).read()
        code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # comments may cite /root/reference file:line
        code = re.sub(r"(//|#).*", "", code)
        code = re.sub(r'""".*?"""', "", code, flags=re.S)
        if re.search(r"\b(from|import)\s+oracle\b|oracle/|liboracle", code):
            bad.append((path, "oracle"))
        if "/root/reference" in code:
            bad.append((path, "reference path in code"))
    assert not bad, bad


def test_bench_and_smoke_use_oracle_only_as_checker():
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # the only oracle import in bench.py lives in cpu_baseline()
    for m in re.finditer(r"from oracle import", bench):
        head = bench[:m.start()]
        assert head.rfind("def cpu_baseline") > head.rfind("def main"), "oracle used outside cpu_baseline()"
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "def smoke" in entry and "def build" in entry


def test_gpu_needed_at_runtime_paths_do_not_read_reference():
    for name in ("bench.py", "__graft_entry__.py"):
        code = re.sub(r"#.*", "", open(os.path.join(ROOT, name)).read())
        code = re.sub(r'""".*?"""', "", code, flags=re.S)
        assert "/root/reference" not in code, name


# ---- build options: the shipped library is built with none (csrc/experiment.hpp) ------------------------------------

CSRC = os.path.join(ROOT, "plonk_gadgets_amd", "csrc")
NOT_OPTIONS = {"PG_EXPERIMENT", "PG_HD"}  # the fence's own key; a function attribute macro


def options_in_conditionals():
    """every PG_... macro a preprocessor conditional of csrc/ looks at"""
    found = set()
    for f in os.listdir(CSRC):
        if f == "experiment.hpp":
            continue
        for line in open(os.path.join(CSRC, f)):
            if re.match(r"\s*#\s*(if|ifdef|ifndef|elif)\b", line):
                found.update(re.findall(r"\bPG_[A-Z0-9_]+\b", line))
    return found - NOT_OPTIONS


def test_the_library_is_built_without_build_options():
    from plonk_gadgets_amd import build as pg_build
    cmd = pg_build.command("/tmp/x.so")
    assert not [a for a in cmd if a.startswith("-D")], cmd
    assert "-O3" in cmd and "--offload-arch=gfx950" in cmd


def test_every_build_option_is_behind_the_experiment_fence():
    fence = open(os.path.join(CSRC, "experiment.hpp")).read()
    fenced = set(re.findall(r"defined\((PG_[A-Z0-9_]+)\)", fence)) - NOT_OPTIONS
    opts = options_in_conditionals()
    assert opts, "no options found: the scan is broken"
    assert opts <= fenced, f"options the sources test but the fence does not name: {sorted(opts - fenced)}"
    assert fenced <= opts, f"the fence names options no source tests any more: {sorted(fenced - opts)}"
    # no build that changes the OUTPUT lives in the sources (they are patches under tools/patches/)
    assert not [o for o in opts if "ABLATE" in o]


def test_an_option_without_the_fence_key_does_not_compile():
    import subprocess
    hdr = os.path.join(CSRC, "experiment.hpp")
    ok = subprocess.run(["g++", "-fsyntax-only", "-x", "c++", hdr], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    for opt in ("-DPG_RC_W=16", "-DPG_MIX_STAMPS", "-DPG_INVERT_FERMAT"):
        bad = subprocess.run(["g++", "-fsyntax-only", "-x", "c++", opt, hdr], capture_output=True, text=True)
        assert bad.returncode != 0 and "PG_EXPERIMENT" in bad.stderr, (opt, bad.stderr)
        good = subprocess.run(["g++", "-fsyntax-only", "-x", "c++", opt, "-DPG_EXPERIMENT", hdr], capture_output=True, text=True)
        assert good.returncode == 0, (opt, good.stderr)
