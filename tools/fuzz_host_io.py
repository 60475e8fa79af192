#!/usr/bin/env python
"""The two host-side look-alikes of the command line against the calls they stand in for, on random and mutated inputs
(no GPU needed):
  write   textio.write_frame  vs  DataFrame.to_csv(quoting=QUOTE_NONNUMERIC)   -- the same bytes, or a refusal (False)
  graphml graphml.read_pathway  vs  relabel_nodes(nx.read_graphml(fp).to_undirected(), attr)  -- the same graph (class, node order,
          typed attributes, adjacency order, graph attributes), or the same exception
    SEED=1 CASES=3000 python tools/fuzz_host_io.py write|graphml"""
import csv
import os
import sys
import tempfile
import warnings

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "write"
rs = np.random.RandomState(int(os.environ.get("SEED", 1)))
ncases = int(os.environ.get("CASES", 1000))
d = tempfile.mkdtemp(prefix="prmf_fuzz_io_")

if what == "write":
    import pandas as pd
    from prmf_amd import textio
    alphabet = list('abcXYZ019 ,;"\'\\\t-_.éß漢\n\r') + ["NA", "nan", ""]
    taken = refused = 0
    for case in range(ncases):
        m, n = int(rs.randint(0, 9)), int(rs.randint(1, 7))
        kind = rs.rand()
        if kind < 0.4:
            a = np.frombuffer(rs.bytes(8 * m * n), dtype=np.float64).reshape(m, n).copy()
        elif kind < 0.7:
            a = (rs.rand(m, n) * 10.0 ** rs.randint(-20, 20, size=(m, n))).astype(np.float32).astype(np.float64)
        else:
            a = rs.choice([0.0, -0.0, 1.0, 1e16, 1e-5, 123456.0, np.inf, -np.inf, np.nan, 5e-324, 1e22, 0.1], size=(m, n))
        if rs.rand() < 0.1:
            a = a.astype(np.float32)
        cols = ["".join(rs.choice(alphabet, size=rs.randint(0, 5))) if rs.rand() < 0.3 else "LV%d" % j for j in range(n)]
        t = rs.rand()
        if t < 0.25:
            index = None
        elif t < 0.5:
            index = ["s%d" % i for i in range(m)]
        elif t < 0.65:
            index = [int(v) for v in rs.randint(-5, 1000, size=m)]
        elif t < 0.9:
            index = ["".join(rs.choice(alphabet, size=rs.randint(0, 6))) for _ in range(m)]
        else:
            index = list(rs.choice([1.5, None, True, "x", 3], size=m)) if m else []
        out, ref = os.path.join(d, "got.csv"), os.path.join(d, "want.csv")
        try:
            ok = textio.write_frame(out, a, index, cols)
        except Exception as e:
            print("CASE %d: write_frame raised %r" % (case, e), repr(index), repr(cols))
            sys.exit(1)
        if not ok:
            refused += 1
            continue
        taken += 1
        pd.DataFrame(a, index=index, columns=cols).to_csv(ref, sep=",", index=index is not None, quoting=csv.QUOTE_NONNUMERIC)
        if open(out, "rb").read() != open(ref, "rb").read():
            print("CASE %d: different bytes" % case, repr(index), repr(cols), a.dtype)
            print(open(out, "rb").read()[:600])
            print(open(ref, "rb").read()[:600])
            sys.exit(1)
    print("%d frames: %d written and byte-identical to pandas, %d refused" % (ncases, taken, refused))
else:
    import networkx as nx
    from prmf_amd import graphml
    from prmf_amd.runner import relabel_nodes
    taken = 0
    for case in range(ncases):
        n = int(rs.randint(1, 25))
        G = nx.Graph()
        for i in rs.permutation(n):
            attrs = {}
            if rs.rand() < 0.85:
                attrs["name"] = "G%d" % (i if rs.rand() < 0.95 else rs.randint(n))        # sometimes two nodes, one name
            if rs.rand() < 0.3:
                attrs["score"] = float(rs.rand())
            if rs.rand() < 0.2:
                attrs["flag"] = bool(rs.rand() < 0.5)
            G.add_node(int(i), **attrs)
        for _ in range(int(rs.randint(0, 3 * n))):
            u, v = int(rs.randint(n)), int(rs.randint(n))
            w = {"weight": float(rs.rand())} if rs.rand() < 0.5 else ({"weight": int(rs.randint(1, 5))} if rs.rand() < 0.3 else {})
            G.add_edge(u, v, **w)
        fp = os.path.join(d, "g.graphml")
        nx.write_graphml(G, fp, named_key_ids=bool(rs.rand() < 0.5))
        text = open(fp, encoding="utf-8").read()
        for _ in range(int(rs.randint(0, 3))):          # mutations of the XML text
            t = rs.randint(10)
            if t == 0:
                text = text.replace('edgedefault="undirected"', 'edgedefault="directed"')
            elif t == 1 and "<edge " in text:
                pos = text.index("<edge ")
                end = text.index("/>", pos) + 2 if "/>" in text[pos:pos + 200] and text.index("/>", pos) < text.index(">", pos) + 1 else text.index("</edge>", pos) + 7
                text = text[:end] + text[pos:end] + text[end:]                     # a parallel edge
            elif t == 2:
                text = text.replace('attr.type="double"', 'attr.type="%s"' % rs.choice(["string", "int", "float", "boolean", "weird"]), 1)
            elif t == 3:
                text = text.replace("</graph>", '<node id="extra"/><edge source="extra" target="nowhere"/></graph>')
            elif t == 4:
                text = text.replace("<data ", "<data><x/></data><data ", 1)
            elif t == 5:
                text = text.replace("</key>", "<default>0</default></key>", 1) if "</key>" in text else text.replace('/>\n', '><default>1</default></key>\n', 1)
            elif t == 6:
                text = text.replace("</graph>", "</graph><graph edgedefault=\"undirected\"><node id=\"z\"/></graph>")
            elif t == 7:
                pos = int(rs.randint(len(text)))
                text = text[:pos] + str(rs.choice(list('<>&"x/ '))) + text[pos + 1:]
            elif t == 8:
                text = text.replace(' id="', ' id="n', 1)
            elif t == 9:
                text = text.replace("<node ", '<node yfiles.foldertype="group" ', 1)
        with open(fp, "w", encoding="utf-8") as fh:
            fh.write(text)
        attr = str(rs.choice(["name", "score", "missing"])) if rs.rand() < 0.9 else None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                want, werr = relabel_nodes(nx.read_graphml(fp).to_undirected(), attr), None
            except BaseException as e:
                want, werr = None, e
            try:
                got, gerr = graphml.read_pathway(fp, attr), None
            except BaseException as e:
                got, gerr = None, e
        try:
            if werr is not None or gerr is not None:
                assert type(werr) is type(gerr) and str(werr) == str(gerr), (werr, gerr)
            else:
                assert type(got) is type(want) and list(got.nodes()) == list(want.nodes())
                assert dict(got.nodes(data=True)) == dict(want.nodes(data=True)) and got.graph == want.graph
                for u in got.nodes():
                    assert list(got.adj[u]) == list(want.adj[u])
                    for v in got.adj[u]:
                        assert got.adj[u][v] == want.adj[u][v]
                        assert {k: type(x) for k, x in got.adj[u][v].items()} == {k: type(x) for k, x in want.adj[u][v].items()}
                assert {k: {a: type(x) for a, x in dd.items()} for k, dd in got.nodes(data=True)} == \
                       {k: {a: type(x) for a, x in dd.items()} for k, dd in want.nodes(data=True)}
                try:
                    graphml._fast(fp, attr)
                    taken += 1
                except graphml._NotPlain:
                    pass
        except AssertionError as e:
            print("CASE %d (attribute %r): %r" % (case, attr, e))
            print(text[:3000])
            sys.exit(1)
    print("%d files: the same graph or the same exception as networkx's three steps; %d took the direct path" % (ncases, taken))
import shutil  # noqa: E402
shutil.rmtree(d, ignore_errors=True)
