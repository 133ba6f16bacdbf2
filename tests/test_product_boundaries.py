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
