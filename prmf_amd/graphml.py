"""Pathway files (GraphML) read once instead of three times.

The reference reads every manifold with `nx.read_graphml(fp).to_undirected()` followed by `prmf.relabel_nodes(G, attr)`
(script/prmf_runner.py:863-866, prmf/__init__.py:65-76): networkx builds a MultiGraph edge by edge, converts it to a Graph,
deep-copies that (`to_undirected`) and copies it again under the new node names -- 10.7 ms per 300-gene pathway, 0.53 s of a
1.5 s run at 10000 x 20000 with 50 pathways, 3 s at 300, 10 s at 1000 (profiles/r05_cli_end_to_end.txt).

`read_pathway` parses the XML with the same ElementTree and builds that final `networkx.Graph` directly: the same class, node
order, node and edge attributes (typed by the file's <key> declarations, edge "id" included), adjacency order and graph
attributes -- `tests/test_host_logic_cpu.py::test_fast_graphml_equals_networkx` compares all of them with the three-step
pipeline.  It takes the plain case only: one undirected <graph>, declared nodes with unique ids, no parallel edges, no
<default> values, hyperedges, ports, nested graphs or yEd extensions, node names (the relabel attribute) that stay unique.
Anything else goes through networkx itself, unchanged -- including every error it raises."""
import xml.etree.ElementTree as ET

NS = "{http://graphml.graphdrawing.org/xmlns}"
_TYPES = {"integer": int, "yfiles": str, "string": str, "int": int, "long": int, "float": float, "double": float, "boolean": bool}
_BOOL = {"true": True, "false": False, "0": False, "1": True}


class _NotPlain(Exception):
    pass


def _slow(fp, node_attribute):
    import networkx as nx
    from .runner import relabel_nodes
    return relabel_nodes(nx.read_graphml(fp).to_undirected(), node_attribute)


def read_pathway(fp, node_attribute="name"):
    """-> the graph `relabel_nodes(nx.read_graphml(fp).to_undirected(), node_attribute)` returns."""
    try:
        return _fast(fp, node_attribute)
    except _NotPlain:
        return _slow(fp, node_attribute)
    except (ET.ParseError, OSError, UnicodeError):
        return _slow(fp, node_attribute)          # let networkx report it in its own words


def _data(elem, keys, tag_data, refused):
    """The <data> children of a node / edge element, typed by their <key>; any other child listed in `refused` (ports, nested
    graphs) makes the file not plain.  One pass over the children."""
    out = {}
    for d in elem:
        tag = d.tag
        if tag != tag_data:
            if tag in refused:
                raise _NotPlain()
            continue
        key = keys.get(d.get("key"))
        if key is None or len(d):
            raise _NotPlain()
        name, typ = key
        text = d.text
        if text is None:
            out[name] = ""
        elif typ is str:
            out[name] = text
        elif typ is bool:
            try:
                out[name] = _BOOL[text.lower()]
            except KeyError:
                raise _NotPlain()
        else:
            try:
                out[name] = typ(text)
            except ValueError:
                raise _NotPlain()
    return out


def _fast(fp, node_attribute):
    import networkx as nx
    if str(fp).lower().endswith((".gz", ".bz2", ".graphmlz")):
        raise _NotPlain()
    root = ET.parse(fp).getroot()
    if root.tag != NS + "graphml":
        raise _NotPlain()
    keys = {}
    for k in root.findall(NS + "key"):
        if k.get("yfiles.type") is not None or k.get("attr.name") is None or k.find(NS + "default") is not None:
            raise _NotPlain()
        typ = _TYPES.get(k.get("attr.type"))           # a missing type: networkx warns and assumes string -- leave it to networkx
        if typ is None:
            raise _NotPlain()
        keys[k.get("id")] = (k.get("attr.name"), typ)
    graphs = root.findall(NS + "graph")
    if len(graphs) != 1:
        raise _NotPlain()
    g = graphs[0]
    if g.get("edgedefault") == "directed":
        raise _NotPlain()
    tag_node, tag_edge, tag_data = NS + "node", NS + "edge", NS + "data"
    refused = (NS + "port", NS + "graph")
    names, node_data, edge_elems = [], {}, []
    for child in g:                                    # one pass: nodes in file order, edges kept for the second step
        tag = child.tag
        if tag == tag_node:
            nid = child.get("id")
            if nid is None or nid in node_data or child.get("yfiles.foldertype") is not None:
                raise _NotPlain()                       # (a yEd group node makes networkx descend into a nested graph)
            node_data[nid] = _data(child, keys, tag_data, refused) if len(child) else {}
            names.append(nid)
        elif tag == tag_edge:
            edge_elems.append(child)
        elif tag == tag_data or tag == NS + "hyperedge":
            raise _NotPlain()                           # graph-level data, hyperedges: networkx's business
    # the relabel: nodes that carry the attribute take its value as their name; the map has to stay one-to-one
    if node_attribute is not None:
        mapping = {n: d[node_attribute] for n, d in node_data.items() if node_attribute in d}
    else:
        mapping = {}
    final = [mapping.get(n, n) for n in names]
    try:
        if len(set(final)) != len(final):
            raise _NotPlain()
    except TypeError:                                   # an unhashable name cannot happen with the types above; be safe
        raise _NotPlain()
    # edges in file order into per-node neighbour lists, as the MultiGraph networkx fills would hold them
    nbrs = {n: {} for n in names}
    for e in edge_elems:
        if e.get("directed") == "true":
            raise _NotPlain()
        s, t = e.get("source"), e.get("target")
        if s not in nbrs or t not in nbrs or t in nbrs[s]:
            raise _NotPlain()                           # undeclared endpoint, or a parallel edge (-> MultiGraph)
        d = _data(e, keys, tag_data, refused) if len(e) else {}
        eid = e.get("id")
        if eid:
            d["id"] = eid
        nbrs[s][t] = d
        nbrs[t][s] = d
    G = nx.Graph()
    G.graph["node_default"] = {}
    G.graph["edge_default"] = {}
    G.add_nodes_from((mapping.get(n, n), node_data[n]) for n in names)
    # Graph(MultiGraph), to_undirected() and relabel_nodes() each re-add the edges node by node, neighbour by neighbour,
    # skipping the ones already seen; the first such pass fixes the adjacency order and the later ones keep it
    seen = set()
    edges = []
    for u in names:
        for v, d in nbrs[u].items():
            if (v, u) in seen:
                continue
            seen.add((u, v))
            edges.append((mapping.get(u, u), mapping.get(v, v), d))
    G.add_edges_from(edges)
    return G
