"""Generates tests/golden/reference_doctests.json: the known-answer examples (statement -> printed answer) that the
reference documents for the sequence path, as golden vectors for tests/test_reference_doctests.py.

Run HERE (the build container, where /root/reference exists); the GPU box only sees the JSON.  Only the
``>>>`` statements and the answers printed under them are taken (``doctest.DocTestParser``) — inputs and expected
outputs, not the prose or the code of the files.  Every source is listed with the SURVEY §8 row it pins; sources of
the reference that document formats / subsystems outside §8 are listed in OUT_OF_SCOPE with the reason, so that
what is not run is stated instead of silently absent.

    python tests/golden/make_reference_doctests.py
"""
import ast
import doctest
import json
import os

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_doctests.json")

# (path in the reference, kind, SURVEY §8 rows it pins)
SOURCES = [
    ("docs_source/topics/kmers.rst", "rst", "A8 A9 A11"),
    ("docs_source/source/encoding.rst", "rst", "T1 A7"),
    ("docs_source/source/reading_files.rst", "rst", "A1-A5 T2"),
    ("docs_source/source/sequences.rst", "rst", "T1 T2"),
    ("README.rst", "rst", "A1-A7"),
    ("bionumpy/io/files.py", "py", "A1 T3"),
    ("bionumpy/sequence/kmers.py", "py", "A8"),
    ("bionumpy/sequence/minimizers.py", "py", "A11"),
    ("bionumpy/sequence/string_matcher.py", "py", "f4"),
    ("bionumpy/sequence/position_weight_matrix.py", "py", "f4"),
    ("bionumpy/streams/decorators.py", "py", "A10"),
    ("bionumpy/encoded_array.py", "py", "T1 T2"),
    ("bionumpy/bnpdataclass/bnpdataclassfunction.py", "py", "T3"),
]

OUT_OF_SCOPE = {
    "docs_source/source/broadcastable_functions.rst": "bnp.arithmetics / intervals (SURVEY §2 rows 20-28)",
    "docs_source/source/intervals.rst": "intervals / bed files",
    "docs_source/source/multiple_data_sources.rst": "genomic_data, bam, vcf",
    "docs_source/source/supported_file_formats.rst": "format table, no examples on the path",
    "docs_source/topics/genome_arithmetics.rst": "genomic_data",
    "docs_source/topics/genomic_data.rst": "genomic_data",
    "docs_source/topics/multiomics.rst": "genomic_data",
    "docs_source/topics/sequence_analysis.rst": "no >>> examples",
    "docs_source/topics/gpu.rst": "the cupy backend switch the north star excludes",
    "bionumpy/sequence/translate.py": "amino-acid translation",
    "bionumpy/streams/groupby_func.py": "groupby over sorted bed/bam streams",
    "bionumpy/streams/multistream.py": "genomic multi-streams",
    "bionumpy/encodings/bool_encoding.py": "bool encoding of vcf flags",
    "bionumpy/io/indexed_bam.py": "bam",
    "bionumpy/io/indexed_files.py": "open_indexed (.fai random access)",
    "bionumpy/bnpdataclass/bnpdataclass.py": "generic @bnpdataclass builder (tables of arbitrary fields)",
}


def _examples(text):
    parser = doctest.DocTestParser()
    names = {v: k for k, v in doctest.OPTIONFLAGS_BY_NAME.items()}
    return [{"source": e.source, "want": e.want, "lineno": e.lineno + 1, "options": {names[k]: v for k, v in e.options.items()}}
            for e in parser.get_examples(text)]


def _docstrings(path):
    """(qualified name, first line, docstring) of every docstring of a python file that holds an example"""
    with open(path) as f:
        src = f.read()
    tree = ast.parse(src)
    out = []

    def visit(node, prefix):
        for child in ast.iter_child_nodes(node):
            if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                name = prefix + child.name
                doc = ast.get_docstring(child, clean=False)
                if doc and ">>>" in doc:
                    body0 = child.body[0]
                    out.append((name, body0.lineno, doc))
                visit(child, name + ".")
    doc = ast.get_docstring(tree, clean=False)
    if doc and ">>>" in doc:
        out.append(("<module>", 1, doc))
    visit(tree, "")
    return out


def main():
    groups = []
    for rel, kind, rows in SOURCES:
        path = os.path.join(REF, rel)
        if kind == "rst":
            with open(path) as f:
                ex = _examples(f.read())
            if ex:
                groups.append({"file": rel, "name": rel, "rows": rows, "first_line": 1, "examples": ex})
        else:
            for name, line, doc in _docstrings(path):
                import textwrap
                ex = _examples(textwrap.dedent(doc))
                for e in ex:
                    e["lineno"] += line - 1
                module = rel[len("bionumpy/"):-len(".py")].replace("/", ".")     # a docstring's examples see their module's names
                groups.append({"file": rel, "name": "%s::%s" % (rel, name), "rows": rows, "first_line": line, "module": module,
                               "examples": ex})
    with open(OUT, "w") as f:
        json.dump({"generated_by": "tests/golden/make_reference_doctests.py", "reference": "bionumpy/bionumpy at /root/reference",
                   "out_of_scope": OUT_OF_SCOPE, "groups": groups}, f, indent=1)
    print("wrote %s: %d groups, %d examples" % (OUT, len(groups), sum(len(g["examples"]) for g in groups)))


if __name__ == "__main__":
    main()
