"""CPU-only: the C-ABI library builds, loads and exports every symbol include/geoa3_hip.h declares."""
import os
import re

from geoa3_amd import _lib

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "geoa3_hip.h")).read()
    declared = set(re.findall(r"\b(geoa3_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.geoa3_version() >= 100
    assert lib.geoa3_strerror(-1) == b"invalid argument"


def test_struct_layouts_match_header():
    """ctypes mirrors have one field per struct member, in order."""
    hdr = open(os.path.join(REPO, "include", "geoa3_hip.h")).read()

    def members(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part.strip())[0])
        return names

    assert members("geoa3_geo_args") == [f[0] for f in _lib.GeoArgs._fields_]
    assert members("geoa3_tnet_weights") == [f[0] for f in _lib.TnetWeights._fields_]
    assert members("geoa3_pointnet_weights") == [f[0] for f in _lib.PointNetWeights._fields_]
    assert members("geoa3_attack_state") == [f[0] for f in _lib.AttackState._fields_]
