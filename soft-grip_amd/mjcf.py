"""MJCF subset compiler: XML -> flat fp64 model (numpy) -> tagged binary blob.

This replaces, for the soft-gripper scenes only, what the reference obtains from
``mujoco_py.load_model_from_path`` (reference environment/manenv.py:27,36).  It
implements exactly the MJCF feature subset listed in SURVEY.md App. A.1:

* ``<include>`` (nested), ``<compiler angle/inertiafromgeom/settotalmass>``,
  ``<option>``, ``<size>``, nested ``<default class=...>``
* bodies, box / capsule / sphere / plane geoms, hinge / slide joints, sites
* spatial tendons through sites (no wrapping geoms / pulleys), fixed tendons
* ``cylinder`` actuators on tendons, ``accelerometer`` / ``gyro`` sensors
* ``<composite type="box"|"ellipsoid">`` shells (radial sliders, joint-fix
  equalities, neighbour equalities between adjacent shell sliders, one fixed
  tendon with a tendon-fix equality)

Section processing order follows MuJoCo's XML reader (all ``<default>`` first,
then every ``<worldbody>`` in document order, then ``<tendon>``, ``<actuator>``,
``<sensor>``).  A composite therefore registers its fixed tendon *before* the
gripper's spatial tendons, i.e. ``OBJT`` is tendon id 0 -- the id the reference
randomises (environment/manenv.py:13,107-108).

Everything MuJoCo-specific here is restated from MuJoCo's documentation; MuJoCo
itself is not available in the build container, so numerical agreement with a
real ``mjModel`` is unpinned (see DESIGN.md "Parity status").
"""
from __future__ import annotations

import os
import struct
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

# ---- enums (MuJoCo numbering, so dumps can be diffed against mj_printModel) ----
JNT_FREE, JNT_SLIDE, JNT_HINGE = 0, 2, 3
GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_BOX = 0, 2, 3, 6
EQ_JOINT, EQ_TENDON = 2, 3
WRAP_JOINT, WRAP_SITE = 1, 3
TRN_TENDON = 3
DYN_FILTER = 2
SENS_ACCELEROMETER, SENS_GYRO = 1, 3

MJ_MINVAL = 1e-15

BLOB_MAGIC = 0x4D474753  # 'SGGM'
BLOB_VERSION = 2


# ----------------------------------------------------------------------------
# small quaternion / rotation helpers (w, x, y, z)
# ----------------------------------------------------------------------------
def _vec(s, n=None, default=None):
    if s is None:
        return None if default is None else np.array(default, dtype=np.float64)
    v = np.array([float(t) for t in s.split()], dtype=np.float64)
    if n is not None and v.size != n:
        raise ValueError("expected %d numbers, got %r" % (n, s))
    return v


def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    n = np.linalg.norm(q)
    if n < MJ_MINVAL:
        return np.array([1.0, 0, 0, 0])
    return q / n


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def quat_z2vec(vec):
    """Minimal rotation taking +z to ``vec`` (MuJoCo's mjuu_z2quat)."""
    vec = np.asarray(vec, dtype=np.float64)
    vec = vec / np.linalg.norm(vec)
    axis = np.cross([0.0, 0.0, 1.0], vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        axis = np.array([1.0, 0.0, 0.0])
    else:
        axis = axis / s
    ang = np.arctan2(s, vec[2])
    return np.concatenate([[np.cos(ang / 2)], axis * np.sin(ang / 2)])


# ----------------------------------------------------------------------------
# XML loading with <include>
# ----------------------------------------------------------------------------
MAX_INCLUDE_DEPTH = 16   # csrc/sg_mjcf.cpp kMaxIncludeDepth


def _load_xml(path: str, base_dir: Optional[str] = None, depth: int = 0) -> ET.Element:
    root = ET.parse(path).getroot()
    if root.tag != "mujoco":
        raise ValueError("%s: root element must be <mujoco>" % path)
    base_dir = base_dir or os.path.dirname(os.path.abspath(path))

    def expand(parent):
        out = []
        for child in list(parent):
            if child.tag == "include":
                if depth >= MAX_INCLUDE_DEPTH:
                    raise ValueError("<include file=%r> nests deeper than %d files (an include cycle?)" % (child.attrib["file"], MAX_INCLUDE_DEPTH))
                inc = _load_xml(os.path.join(base_dir, child.attrib["file"]), base_dir, depth + 1)
                out.extend(list(inc))
            else:
                expand(child)
                out.append(child)
        for c in list(parent):
            parent.remove(c)
        for c in out:
            parent.append(c)

    expand(root)
    return root


# ----------------------------------------------------------------------------
# defaults
# ----------------------------------------------------------------------------
class _Defaults:
    def __init__(self):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}
        self.parent: Dict[str, Optional[str]] = {"main": None}

    def read(self, elem: ET.Element, cls: str, parent: Optional[str]):
        if cls not in self.classes:
            self.classes[cls] = {}
            self.parent[cls] = parent
            if parent is not None:  # inherit a copy of the parent's settings
                for tag, attrs in self.classes[parent].items():
                    self.classes[cls][tag] = dict(attrs)
        for child in elem:  # own settings first, nested classes (which inherit them) second
            if child.tag != "default":
                self.classes[cls].setdefault(child.tag, {}).update(child.attrib)
        for child in elem:
            if child.tag == "default":
                self.read(child, child.attrib["class"], cls)

    def resolve(self, tag: str, elem_attrib: Dict[str, str], childclass: Optional[str] = None):
        cls = elem_attrib.get("class", childclass or "main")
        if cls not in self.classes:
            raise ValueError("unknown default class %r" % cls)
        out = dict(self.classes[cls].get(tag, {}))
        out.update(elem_attrib)
        return out


# ----------------------------------------------------------------------------
# intermediate spec objects
# ----------------------------------------------------------------------------
@dataclass
class _Geom:
    name: str
    type: int
    size: np.ndarray
    pos: np.ndarray
    quat: np.ndarray
    mass: Optional[float]
    density: float
    contype: int
    conaffinity: int
    condim: int
    friction: np.ndarray
    solref: np.ndarray
    solimp: np.ndarray
    solmix: float
    margin: float
    gap: float
    priority: int


@dataclass
class _Joint:
    name: str
    type: int
    pos: np.ndarray
    axis: np.ndarray
    limited: bool
    range: np.ndarray
    stiffness: float
    damping: float
    armature: float
    margin: float
    ref: float
    springref: float
    solreflimit: np.ndarray
    solimplimit: np.ndarray


@dataclass
class _Site:
    name: str
    pos: np.ndarray
    quat: np.ndarray


@dataclass
class _Body:
    name: str
    pos: np.ndarray
    quat: np.ndarray
    parent: int
    geoms: List[_Geom] = field(default_factory=list)
    joints: List[_Joint] = field(default_factory=list)
    sites: List[_Site] = field(default_factory=list)


@dataclass
class _Tendon:
    name: str
    kind: str  # "spatial" | "fixed"
    wraps: list  # spatial: [site names]; fixed: [(joint name, coef)]
    stiffness: float
    damping: float
    springlength: float


@dataclass
class _Equality:
    type: int
    name1: str
    solref: np.ndarray
    solimp: np.ndarray
    data: np.ndarray
    name2: Optional[str] = None  # second joint of a two-joint equality (q1 - q1_0 = poly(q2 - q2_0), data = polycoef)


_DEF_SOLREF = np.array([0.02, 1.0])
_DEF_SOLIMP = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
_GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE, "box": GEOM_BOX}


def _parse_solimp(s):
    v = _vec(s)
    out = _DEF_SOLIMP.copy()
    out[: v.size] = v
    return out


def _bool(s, default=False):
    if s is None:
        return default
    return s.strip().lower() == "true"


class _Compiler:
    def __init__(self, path: str, composite_neighbors: bool = True):
        self.root = _load_xml(path)
        self.composite_neighbors = composite_neighbors
        self.defaults = _Defaults()
        self.bodies: List[_Body] = [_Body("world", np.zeros(3), np.array([1.0, 0, 0, 0]), -1)]
        self.tendons: List[_Tendon] = []
        self.equalities: List[_Equality] = []
        self.actuators: list = []
        self.sensors: list = []
        self.settotalmass = -1.0
        self.opt = dict(timestep=0.002, gravity=np.array([0, 0, -9.81]), iterations=100,
                        tolerance=1e-8, impratio=1.0, solver="Newton", cone="pyramidal")
        self.size = dict(nconmax=-1, njmax=-1)

    # -- sections -----------------------------------------------------------
    def run(self):
        r = self.root
        for e in r.findall("compiler"):
            if e.attrib.get("angle", "degree") != "radian":
                raise ValueError("only angle='radian' is supported")
            if "settotalmass" in e.attrib:
                self.settotalmass = float(e.attrib["settotalmass"])
        for e in r.findall("option"):
            a = e.attrib
            if "timestep" in a:
                self.opt["timestep"] = float(a["timestep"])
            if "gravity" in a:
                self.opt["gravity"] = _vec(a["gravity"], 3)
            if "iterations" in a:
                self.opt["iterations"] = int(a["iterations"])
            if "tolerance" in a:
                self.opt["tolerance"] = float(a["tolerance"])
            if "impratio" in a:
                self.opt["impratio"] = float(a["impratio"])
            if "solver" in a:
                self.opt["solver"] = a["solver"]
            if "cone" in a:
                self.opt["cone"] = a["cone"]
        for e in r.findall("size"):
            for k in ("nconmax", "njmax"):
                if k in e.attrib:
                    self.size[k] = int(e.attrib[k])
        for e in r.findall("default"):
            self.defaults.read(e, "main", None)
        for e in r.findall("worldbody"):
            self._body_children(e, 0, None)
        for e in r.findall("tendon"):
            self._tendon_section(e)
        for e in r.findall("actuator"):
            for a in e:
                if a.tag != "cylinder":
                    raise ValueError("unsupported actuator <%s>" % a.tag)
                at = self.defaults.resolve("cylinder", a.attrib)
                if "tendon" not in at:
                    raise ValueError("cylinder actuators must act on a tendon")
                area = float(at.get("area", 1.0))
                if "diameter" in at:
                    area = np.pi * float(at["diameter"]) ** 2 / 4
                bias = _vec(at.get("bias"), 3, default=[0, 0, 0])
                self.actuators.append(dict(tendon=at["tendon"], timeconst=float(at.get("timeconst", 1.0)),
                                           gain=area, bias=bias, gear=float(at.get("gear", "1").split()[0])))
        for e in r.findall("sensor"):
            for s in e:
                if s.tag not in ("accelerometer", "gyro"):
                    raise ValueError("unsupported sensor <%s>" % s.tag)
                self.sensors.append(dict(type=SENS_ACCELEROMETER if s.tag == "accelerometer" else SENS_GYRO,
                                         site=s.attrib["site"], name=s.attrib.get("name", "")))
        if self.opt["solver"] != "PGS" or self.opt["cone"] != "elliptic":
            raise ValueError("only solver='PGS' cone='elliptic' (reference soft_scene.xml:13) is implemented")
        return self._finalize()

    # -- worldbody ----------------------------------------------------------
    def _make_geom(self, at: Dict[str, str], name: str) -> _Geom:
        gtype = _GEOM_TYPES.get(at.get("type", "sphere"))
        if gtype is None:
            raise ValueError("unsupported geom type %r" % at.get("type"))
        size = np.zeros(3)
        sv = _vec(at.get("size"), default=[0, 0, 0])
        size[: sv.size] = sv
        fr = np.array([1.0, 0.005, 0.0001])
        if "friction" in at:
            fv = _vec(at["friction"])
            fr[: fv.size] = fv
        return _Geom(
            name=name, type=gtype, size=size,
            pos=_vec(at.get("pos"), 3, default=[0, 0, 0]),
            quat=quat_normalize(_vec(at.get("quat"), 4, default=[1, 0, 0, 0])),
            mass=float(at["mass"]) if "mass" in at else None,
            density=float(at.get("density", 1000.0)),
            contype=int(at.get("contype", 1)), conaffinity=int(at.get("conaffinity", 1)),
            condim=int(at.get("condim", 3)), friction=fr,
            solref=_vec(at.get("solref"), 2, default=_DEF_SOLREF),
            solimp=_parse_solimp(at["solimp"]) if "solimp" in at else _DEF_SOLIMP.copy(),
            solmix=float(at.get("solmix", 1.0)), margin=float(at.get("margin", 0.0)),
            gap=float(at.get("gap", 0.0)), priority=int(at.get("priority", 0)))

    def _make_joint(self, at: Dict[str, str], name: str) -> _Joint:
        t = at.get("type", "hinge")
        if t not in ("hinge", "slide"):
            raise ValueError("unsupported joint type %r" % t)
        axis = _vec(at.get("axis"), 3, default=[0, 0, 1])
        axis = axis / np.linalg.norm(axis)
        return _Joint(
            name=name, type=JNT_HINGE if t == "hinge" else JNT_SLIDE,
            pos=_vec(at.get("pos"), 3, default=[0, 0, 0]), axis=axis,
            limited=_bool(at.get("limited")), range=_vec(at.get("range"), 2, default=[0, 0]),
            stiffness=float(at.get("stiffness", 0.0)), damping=float(at.get("damping", 0.0)),
            armature=float(at.get("armature", 0.0)), margin=float(at.get("margin", 0.0)),
            ref=float(at.get("ref", 0.0)), springref=float(at.get("springref", 0.0)),
            solreflimit=_vec(at.get("solreflimit"), 2, default=_DEF_SOLREF),
            solimplimit=_parse_solimp(at["solimplimit"]) if "solimplimit" in at else _DEF_SOLIMP.copy())

    def _body_children(self, elem: ET.Element, bid: int, childclass: Optional[str]):
        body = self.bodies[bid]
        for c in elem:
            if c.tag == "geom":
                at = self.defaults.resolve("geom", c.attrib, childclass)
                body.geoms.append(self._make_geom(at, c.attrib.get("name", "")))
            elif c.tag == "joint":
                at = self.defaults.resolve("joint", c.attrib, childclass)
                body.joints.append(self._make_joint(at, c.attrib.get("name", "")))
            elif c.tag == "site":
                at = self.defaults.resolve("site", c.attrib, childclass)
                body.sites.append(_Site(c.attrib.get("name", ""), _vec(at.get("pos"), 3, default=[0, 0, 0]),
                                        quat_normalize(_vec(at.get("quat"), 4, default=[1, 0, 0, 0]))))
            elif c.tag == "body":
                nb = _Body(c.attrib.get("name", ""), _vec(c.attrib.get("pos"), 3, default=[0, 0, 0]),
                           quat_normalize(_vec(c.attrib.get("quat"), 4, default=[1, 0, 0, 0])), bid)
                self.bodies.append(nb)
                self._body_children(c, len(self.bodies) - 1, c.attrib.get("childclass", childclass))
            elif c.tag == "composite":
                self._composite(c, bid, childclass)
            elif c.tag in ("light", "camera", "inertial"):
                if c.tag == "inertial":
                    raise ValueError("<inertial> is not supported (inertiafromgeom only)")
            elif c.tag == "freejoint":
                # SURVEY.md 8(f) rank 4 (reference data/gripper/soft_experiments_softball.xml:8): 7 positions (world position +
                # quaternion), 6 dofs (linear velocity in the world frame, angular velocity in the body frame); no spring, damper,
                # armature or limit.  Compiled here and run by the oracle; the kernels' plans refuse it with a reason.
                if body.parent != 0 or body.joints:
                    raise ValueError("a free joint must be the only joint of a child of the world body")
                body.joints.append(_Joint(
                    name=c.attrib.get("name", ""), type=JNT_FREE, pos=np.zeros(3), axis=np.array([0.0, 0.0, 1.0]), limited=False,
                    range=np.zeros(2), stiffness=0.0, damping=0.0, armature=0.0, margin=0.0, ref=0.0, springref=0.0,
                    solreflimit=_DEF_SOLREF.copy(), solimplimit=_DEF_SOLIMP.copy()))
            else:
                raise ValueError("unsupported worldbody element <%s>" % c.tag)

    # -- composite -------------------------------------
    def _composite(self, elem: ET.Element, bid: int, childclass: Optional[str]):
        ctype = elem.attrib["type"]
        if ctype not in ("box", "ellipsoid", "cylinder"):
            raise ValueError("unsupported composite type %r" % ctype)
        prefix = elem.attrib.get("prefix", "")
        fcount = [float(t) for t in elem.attrib["count"].split()]
        if len(fcount) != 3:
            raise ValueError("box/ellipsoid composites need a 3-D count")
        if any(not (2 <= c <= 64) or c != int(c) for c in fcount):
            raise ValueError("composite count must be whole numbers in [2, 64] per axis, got %r" % elem.attrib["count"])
        count = [int(c) for c in fcount]
        shell = count[0] * count[1] * count[2] - max(0, (count[0] - 2) * (count[1] - 2) * (count[2] - 2))
        if shell > 256:
            raise ValueError("composite with %d shell elements: at most 256 are supported" % shell)
        spacing = float(elem.attrib["spacing"])
        gattr = self.defaults.resolve("geom", {}, childclass)
        jattr = self.defaults.resolve("joint", {}, childclass)
        tattr: Dict[str, str] = {}
        eq_j = dict(solref=_DEF_SOLREF.copy(), solimp=_DEF_SOLIMP.copy())
        eq_t = dict(solref=_DEF_SOLREF.copy(), solimp=_DEF_SOLIMP.copy())
        for c in elem:
            if c.tag == "geom":
                gattr.update(c.attrib)
            elif c.tag == "joint":
                if c.attrib.get("kind", "main") != "main":
                    raise ValueError("only <joint kind='main'> is supported in composites")
                a = dict(c.attrib)
                if "solreffix" in a:
                    eq_j["solref"] = _vec(a.pop("solreffix"), 2)
                if "solimpfix" in a:
                    eq_j["solimp"] = _parse_solimp(a.pop("solimpfix"))
                a.pop("kind", None)
                jattr.update(a)
            elif c.tag == "tendon":
                a = dict(c.attrib)
                if "solreffix" in a:
                    eq_t["solref"] = _vec(a.pop("solreffix"), 2)
                if "solimpfix" in a:
                    eq_t["solimp"] = _parse_solimp(a.pop("solimpfix"))
                a.pop("kind", None)
                tattr.update(a)
            elif c.tag == "skin":
                pass  # render-only
            else:
                raise ValueError("unsupported composite child <%s>" % c.tag)

        parent = self.bodies[bid]
        gc = self._make_geom(gattr, prefix + "Gcenter")
        gc.type = GEOM_SPHERE
        gc.pos = np.zeros(3)
        gc.size = np.array([gc.size[0] * 2, 0.0, 0.0])
        parent.geoms.append(gc)

        half = [0.5 * spacing * (n - 1) for n in count]
        ten = _Tendon(prefix + "T", "fixed", [], float(tattr.get("stiffness", 0.0)),
                      float(tattr.get("damping", 0.0)), -1.0)
        self.tendons.append(ten)
        for ix in range(count[0]):
            for iy in range(count[1]):
                for iz in range(count[2]):
                    if not (ix in (0, count[0] - 1) or iy in (0, count[1] - 1) or iz in (0, count[2] - 1)):
                        continue
                    p = np.array([2.0 * ix / (count[0] - 1) - 1, 2.0 * iy / (count[1] - 1) - 1,
                                  2.0 * iz / (count[2] - 1) - 1])
                    if ctype == "box":
                        p = p * half
                    elif ctype == "ellipsoid":
                        p = p / np.linalg.norm(p) * half
                    else:  # cylinder
                        l0 = max(abs(p[0]), abs(p[1]))
                        n2 = np.linalg.norm(p[:2])
                        if n2 < MJ_MINVAL:  # odd counts put an element on the axis (the centre of a cap): it stays there
                            p = np.array([0.0, 0.0, p[2] * half[2]])
                        else:
                            p = np.array([p[0] / n2 * half[0] * l0, p[1] / n2 * half[1] * l0, p[2] * half[2]])
                    tag = "%d_%d_%d" % (ix, iy, iz)
                    b = _Body(prefix + "B" + tag, p, quat_z2vec(p), bid)
                    g = self._make_geom(gattr, prefix + "G" + tag)
                    if g.type == GEOM_CAPSULE:
                        g.pos = np.array([0.0, 0.0, -(g.size[0] + g.size[1])])
                    else:
                        g.type = GEOM_SPHERE
                        g.pos = np.array([0.0, 0.0, -g.size[0]])
                    b.geoms.append(g)
                    ja = dict(jattr)
                    ja.update(type="slide", pos="0 0 0", axis="0 0 1")
                    j = self._make_joint(ja, prefix + "J" + tag)
                    b.joints.append(j)
                    self.bodies.append(b)
                    self.equalities.append(_Equality(EQ_JOINT, j.name, eq_j["solref"], eq_j["solimp"], np.zeros(5)))
                    ten.wraps.append((j.name, 1.0))
                    if self.composite_neighbors:
                        # "each joint is equality-constrained to remain equal to its neighbor joints" (MuJoCo 2.x composite
                        # documentation, box / cylinder / ellipsoid): one two-joint equality towards the next shell element
                        # along +x, +y, +z, registered right after the element's own fix row, same solreffix / solimpfix
                        for d in range(3):
                            q = [ix, iy, iz]
                            q[d] = min(q[d] + 1, count[d] - 1)
                            if q == [ix, iy, iz] or not any(q[k] in (0, count[k] - 1) for k in range(3)):
                                continue
                            self.equalities.append(_Equality(EQ_JOINT, j.name, eq_j["solref"], eq_j["solimp"],
                                                             np.array([0.0, 1.0, 0.0, 0.0, 0.0]),
                                                             prefix + "J%d_%d_%d" % tuple(q)))
        self.equalities.append(_Equality(EQ_TENDON, ten.name, eq_t["solref"], eq_t["solimp"], np.zeros(5)))

    # -- tendons ------------------------------------------------------------
    def _tendon_section(self, elem: ET.Element):
        for t in elem:
            at = self.defaults.resolve("tendon", t.attrib)
            if t.tag == "spatial":
                wraps = []
                for w in t:
                    if w.tag != "site":
                        raise ValueError("only site wraps are supported in spatial tendons")
                    wraps.append(w.attrib["site"])
                if len(wraps) < 2:
                    raise ValueError("spatial tendon needs >= 2 sites")
                self.tendons.append(_Tendon(t.attrib.get("name", ""), "spatial", wraps,
                                            float(at.get("stiffness", 0.0)), float(at.get("damping", 0.0)),
                                            float(at.get("springlength", -1.0))))
            elif t.tag == "fixed":
                wraps = [(w.attrib["joint"], float(w.attrib["coef"])) for w in t]
                self.tendons.append(_Tendon(t.attrib.get("name", ""), "fixed", wraps,
                                            float(at.get("stiffness", 0.0)), float(at.get("damping", 0.0)),
                                            float(at.get("springlength", -1.0))))
            else:
                raise ValueError("unsupported tendon <%s>" % t.tag)

    # -- finalize: flatten to arrays -----------------------------------------
    def _finalize(self) -> "Model":
        B = self.bodies
        nbody = len(B)
        m = Model()
        m.opt_timestep = self.opt["timestep"]
        m.opt_gravity = np.asarray(self.opt["gravity"], dtype=np.float64)
        m.opt_iterations = self.opt["iterations"]
        m.opt_tolerance = self.opt["tolerance"]
        m.opt_impratio = self.opt["impratio"]
        m.opt_implicit_tendon_damping = 0   # DESIGN.md D5; load_model(..., tendon_damper="implicit") sets it
        m.nconmax = self.size["nconmax"]
        m.njmax = self.size["njmax"]

        m.body_names = [b.name for b in B]
        m.body_parentid = np.array([max(b.parent, 0) for b in B], dtype=np.int32)
        m.body_pos = np.array([b.pos for b in B])
        m.body_quat = np.array([b.quat for b in B])
        jn, gn, sn = [], [], []
        m.body_jntadr = np.zeros(nbody, np.int32)
        m.body_jntnum = np.zeros(nbody, np.int32)
        m.body_geomadr = np.zeros(nbody, np.int32)
        m.body_geomnum = np.zeros(nbody, np.int32)
        joints, geoms, sites = [], [], []
        for i, b in enumerate(B):
            m.body_jntadr[i] = len(joints) if b.joints else -1
            m.body_jntnum[i] = len(b.joints)
            m.body_geomadr[i] = len(geoms) if b.geoms else -1
            m.body_geomnum[i] = len(b.geoms)
            for j in b.joints:
                joints.append((i, j))
            for g in b.geoms:
                geoms.append((i, g))
            for s in b.sites:
                sites.append((i, s))
        nj, ng, ns = len(joints), len(geoms), len(sites)
        m.jnt_names = [j.name for _, j in joints]
        m.geom_names = [g.name for _, g in geoms]
        m.site_names = [s.name for _, s in sites]

        # weld ids: first ancestor-or-self with a joint; 0 when rigidly attached to the world
        m.body_weldid = np.zeros(nbody, np.int32)
        for i in range(1, nbody):
            m.body_weldid[i] = i if B[i].joints else m.body_weldid[B[i].parent]

        m.jnt_type = np.array([j.type for _, j in joints], np.int32)
        m.jnt_bodyid = np.array([i for i, _ in joints], np.int32)
        m.jnt_pos = np.array([j.pos for _, j in joints]).reshape(nj, 3)
        m.jnt_axis = np.array([j.axis for _, j in joints]).reshape(nj, 3)
        m.jnt_limited = np.array([int(j.limited) for _, j in joints], np.int32)
        m.jnt_range = np.array([j.range for _, j in joints]).reshape(nj, 2)
        m.jnt_stiffness = np.array([j.stiffness for _, j in joints])
        m.jnt_margin = np.array([j.margin for _, j in joints])
        m.jnt_solref = np.array([j.solreflimit for _, j in joints]).reshape(nj, 2)
        m.jnt_solimp = np.array([j.solimplimit for _, j in joints]).reshape(nj, 5)
        # positions (nq) and dofs (nv): one each for a hinge / slide; a free joint has 7 positions (the body's world position and
        # quaternion: a child of the world) and 6 dofs
        q0, qs, dd, da, qadr, dadr, djnt = [], [], [], [], [], [], []
        for k, (i, j) in enumerate(joints):
            qadr.append(len(q0)); dadr.append(len(dd))
            if j.type == JNT_FREE:
                pose = list(B[i].pos) + list(B[i].quat)
                q0 += pose; qs += pose
                dd += [j.damping] * 6; da += [j.armature] * 6; djnt += [k] * 6
            else:
                q0.append(j.ref); qs.append(j.springref); dd.append(j.damping); da.append(j.armature); djnt.append(k)
        m.qpos0, m.qpos_spring = np.array(q0, dtype=np.float64), np.array(qs, dtype=np.float64)
        m.dof_damping, m.dof_armature = np.array(dd, dtype=np.float64), np.array(da, dtype=np.float64)
        m.jnt_qposadr, m.jnt_dofadr, m.dof_jntid = np.array(qadr, np.int32), np.array(dadr, np.int32), np.array(djnt, np.int32)
        nvd = len(dd)
        # dof tree: previous dof on the same body, else last dof of the nearest jointed ancestor
        m.dof_parentid = np.full(nvd, -1, np.int32)
        last_dof_of_body = np.full(nbody, -1, np.int32)
        for i in range(1, nbody):
            prev = last_dof_of_body[B[i].parent]
            for k in range(m.body_jntnum[i]):
                jj = m.body_jntadr[i] + k
                for d in range(dadr[jj], dadr[jj] + (6 if joints[jj][1].type == JNT_FREE else 1)):
                    m.dof_parentid[d] = prev
                    prev = d
            last_dof_of_body[i] = prev

        m.geom_type = np.array([g.type for _, g in geoms], np.int32)
        m.geom_bodyid = np.array([i for i, _ in geoms], np.int32)
        m.geom_contype = np.array([g.contype for _, g in geoms], np.int32)
        m.geom_conaffinity = np.array([g.conaffinity for _, g in geoms], np.int32)
        m.geom_condim = np.array([g.condim for _, g in geoms], np.int32)
        m.geom_priority = np.array([g.priority for _, g in geoms], np.int32)
        m.geom_size = np.array([g.size for _, g in geoms]).reshape(ng, 3)
        m.geom_pos = np.array([g.pos for _, g in geoms]).reshape(ng, 3)
        m.geom_quat = np.array([g.quat for _, g in geoms]).reshape(ng, 4)
        m.geom_friction = np.array([g.friction for _, g in geoms]).reshape(ng, 3)
        m.geom_solref = np.array([g.solref for _, g in geoms]).reshape(ng, 2)
        m.geom_solimp = np.array([g.solimp for _, g in geoms]).reshape(ng, 5)
        m.geom_solmix = np.array([g.solmix for _, g in geoms])
        m.geom_margin = np.array([g.margin for _, g in geoms])
        m.geom_gap = np.array([g.gap for _, g in geoms])
        rb = np.zeros(ng)
        for k, (_, g) in enumerate(geoms):
            if g.type == GEOM_SPHERE:
                rb[k] = g.size[0]
            elif g.type == GEOM_CAPSULE:
                rb[k] = g.size[0] + g.size[1]
            elif g.type == GEOM_BOX:
                rb[k] = np.linalg.norm(g.size)
        m.geom_rbound = rb

        m.site_bodyid = np.array([i for i, _ in sites], np.int32)
        m.site_pos = np.array([s.pos for _, s in sites]).reshape(ns, 3)
        m.site_quat = np.array([s.quat for _, s in sites]).reshape(ns, 4)

        # ---- inertial properties from geoms (inertiafromgeom) ----
        mass = np.zeros(nbody)
        ipos = np.zeros((nbody, 3))
        imat = np.zeros((nbody, 3, 3))  # inertia about the COM, body frame
        for i, b in enumerate(B):
            gm, gI = [], []
            for g in b.geoms:
                vol, Id = _geom_volume_inertia(g)
                gmass = g.mass if g.mass is not None else g.density * vol
                gm.append(gmass)
                gI.append(Id * gmass)  # Id is per unit mass
            if not gm or sum(gm) <= 0:
                continue
            mass[i] = sum(gm)
            ipos[i] = sum(mm * g.pos for mm, g in zip(gm, b.geoms)) / mass[i]
            for mm, Idiag, g in zip(gm, gI, b.geoms):
                R = quat_to_mat(g.quat)
                d = g.pos - ipos[i]
                imat[i] += R @ np.diag(Idiag) @ R.T + mm * (d @ d * np.eye(3) - np.outer(d, d))
        if self.settotalmass > 0:
            scale = self.settotalmass / max(MJ_MINVAL, mass[1:].sum())
            mass *= scale
            imat *= scale
        m.body_mass, m.body_ipos, m.body_imat = mass, ipos, imat.reshape(nbody, 9)
        for i in range(1, nbody):
            if m.body_weldid[i] != 0 and m.body_jntnum[i] > 0 and mass[i] < MJ_MINVAL:
                raise ValueError("moving body %d (%s) has no mass" % (i, B[i].name))

        # ---- tendons / equality / actuators / sensors ----
        jidx = {n: k for k, n in enumerate(m.jnt_names) if n}
        sidx = {n: k for k, n in enumerate(m.site_names) if n}
        wt, wo, wp, tadr, tnum = [], [], [], [], []
        for t in self.tendons:
            tadr.append(len(wt))
            if t.kind == "spatial":
                for s in t.wraps:
                    wt.append(WRAP_SITE)
                    wo.append(sidx[s])
                    wp.append(0.0)
            else:
                for jname, coef in t.wraps:
                    wt.append(WRAP_JOINT)
                    wo.append(jidx[jname])
                    wp.append(coef)
            tnum.append(len(wt) - tadr[-1])
        nt = len(self.tendons)
        m.tendon_names = [t.name for t in self.tendons]
        m.tendon_adr = np.array(tadr, np.int32).reshape(nt)
        m.tendon_num = np.array(tnum, np.int32).reshape(nt)
        m.tendon_stiffness = np.array([t.stiffness for t in self.tendons]).reshape(nt)
        m.tendon_damping = np.array([t.damping for t in self.tendons]).reshape(nt)
        m.wrap_type = np.array(wt, np.int32)
        m.wrap_objid = np.array(wo, np.int32)
        m.wrap_prm = np.array(wp, np.float64)

        tidx = {n: k for k, n in enumerate(m.tendon_names) if n}
        ne = len(self.equalities)
        m.eq_type = np.array([e.type for e in self.equalities], np.int32).reshape(ne)
        m.eq_obj1id = np.array([jidx[e.name1] if e.type == EQ_JOINT else tidx[e.name1]
                                for e in self.equalities], np.int32).reshape(ne)
        m.eq_obj2id = np.array([jidx[e.name2] if e.name2 is not None else -1 for e in self.equalities],
                               np.int32).reshape(ne)
        m.eq_solref = np.array([e.solref for e in self.equalities]).reshape(ne, 2)
        m.eq_solimp = np.array([e.solimp for e in self.equalities]).reshape(ne, 5)
        m.eq_data = np.array([e.data for e in self.equalities]).reshape(ne, 5)

        nu = len(self.actuators)
        m.actuator_trnid = np.array([tidx[a["tendon"]] for a in self.actuators], np.int32).reshape(nu)
        m.actuator_timeconst = np.array([a["timeconst"] for a in self.actuators]).reshape(nu)
        m.actuator_gain = np.array([a["gain"] for a in self.actuators]).reshape(nu)
        m.actuator_bias = np.array([a["bias"] for a in self.actuators]).reshape(nu, 3)
        m.actuator_gear = np.array([a["gear"] for a in self.actuators]).reshape(nu)

        nsens = len(self.sensors)
        m.sensor_type = np.array([s["type"] for s in self.sensors], np.int32).reshape(nsens)
        m.sensor_objid = np.array([sidx[s["site"]] for s in self.sensors], np.int32).reshape(nsens)
        m.sensor_adr = np.arange(nsens, dtype=np.int32) * 3
        m.sensor_names = [s["name"] for s in self.sensors]

        m._set_const()
        return m


def _geom_volume_inertia(g: _Geom):
    """Volume and per-unit-mass principal inertia in the geom frame (MuJoCo's shape formulas)."""
    if g.type == GEOM_SPHERE:
        r = g.size[0]
        return 4.0 / 3.0 * np.pi * r ** 3, np.full(3, 0.4 * r * r)
    if g.type == GEOM_BOX:
        sx, sy, sz = g.size
        return 8 * sx * sy * sz, np.array([sy * sy + sz * sz, sx * sx + sz * sz, sx * sx + sy * sy]) / 3.0
    if g.type == GEOM_CAPSULE:
        r, h = g.size[0], 2 * g.size[1]
        vol = np.pi * (r * r * h + 4.0 / 3.0 * r ** 3)
        ms = 4 * r / (4 * r + 3 * h)  # mass fraction of the two hemispheres
        mc = 1.0 - ms
        ixy = mc * (3 * r * r + h * h) / 12 + 2 * ms * r * r / 5 + ms * h * (3 * r + 2 * h) / 8
        iz = mc * r * r / 2 + 2 * ms * r * r / 5
        return vol, np.array([ixy, ixy, iz])
    if g.type == GEOM_PLANE:
        return 0.0, np.zeros(3)
    raise ValueError("geom type %d" % g.type)


# ----------------------------------------------------------------------------
# compiled model
# ----------------------------------------------------------------------------
class Model:
    """Flat fp64 model; field names follow mjModel where a counterpart exists."""

    # ---- kinematics at an arbitrary qpos (host-side, used for set_const and tests) ----
    def kinematics(self, qpos):
        nb = len(self.body_parentid)
        xpos = np.zeros((nb, 3))
        xquat = np.zeros((nb, 4))
        xquat[0] = [1, 0, 0, 0]
        nj = len(self.jnt_type)
        xanchor = np.zeros((nj, 3))
        xaxis = np.zeros((nj, 3))
        qa = self.jnt_qposadr
        for i in range(1, nb):
            p = self.body_parentid[i]
            R = quat_to_mat(xquat[p])
            pos = xpos[p] + R @ self.body_pos[i]
            quat = quat_mul(xquat[p], self.body_quat[i])
            for k in range(self.body_jntnum[i]):
                j = self.body_jntadr[i] + k
                if self.jnt_type[j] == JNT_FREE:     # the pose IS the joint's 7 positions
                    pos = np.array(qpos[qa[j]:qa[j] + 3], dtype=np.float64)
                    quat = quat_normalize(np.array(qpos[qa[j] + 3:qa[j] + 7], dtype=np.float64))
                    xanchor[j] = pos
                    continue
                R = quat_to_mat(quat)
                xanchor[j] = pos + R @ self.jnt_pos[j]
                xaxis[j] = R @ self.jnt_axis[j]
                dq = qpos[qa[j]] - self.qpos0[qa[j]]
                if self.jnt_type[j] == JNT_SLIDE:
                    pos = pos + xaxis[j] * dq
                else:
                    ql = np.concatenate([[np.cos(dq / 2)], self.jnt_axis[j] * np.sin(dq / 2)])
                    quat = quat_mul(quat, ql)
                    pos = xanchor[j] - quat_to_mat(quat) @ self.jnt_pos[j]
            xpos[i] = pos
            xquat[i] = quat_normalize(quat)
        xmat = np.array([quat_to_mat(q) for q in xquat])
        return dict(xpos=xpos, xquat=xquat, xmat=xmat, xanchor=xanchor, xaxis=xaxis)

    def _jac_point(self, kin, body, point):
        """3 x nv translational and rotational Jacobians of ``point`` fixed to ``body``."""
        nv = self.nv
        jp = np.zeros((3, nv))
        jr = np.zeros((3, nv))
        da = self.jnt_dofadr
        b = body
        while b > 0:
            for k in range(self.body_jntnum[b]):
                j = self.body_jntadr[b] + k
                d = da[j]
                ax = kin["xaxis"][j]
                if self.jnt_type[j] == JNT_FREE:    # translations along the world axes, rotations about the body's own axes
                    R = kin["xmat"][b]
                    for c in range(3):
                        jp[c, d + c] = 1.0
                        jr[:, d + 3 + c] = R[:, c]
                        jp[:, d + 3 + c] = np.cross(R[:, c], point - kin["xpos"][b])
                elif self.jnt_type[j] == JNT_SLIDE:
                    jp[:, d] = ax
                else:
                    jr[:, d] = ax
                    jp[:, d] = np.cross(ax, point - kin["xanchor"][j])
            b = self.body_parentid[b]
        return jp, jr

    def mass_matrix(self, qpos):
        kin = self.kinematics(qpos)
        M = np.diag(self.dof_armature.astype(np.float64))
        for b in range(1, len(self.body_parentid)):
            if self.body_mass[b] <= 0 or self.body_weldid[b] == 0:
                continue
            R = kin["xmat"][b]
            com = kin["xpos"][b] + R @ self.body_ipos[b]
            Iw = R @ self.body_imat[b].reshape(3, 3) @ R.T
            jp, jr = self._jac_point(kin, b, com)
            M += self.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
        return M, kin

    def site_xpos(self, kin):
        return np.array([kin["xpos"][b] + kin["xmat"][b] @ p for b, p in zip(self.site_bodyid, self.site_pos)])

    def tendon_length_jac(self, qpos, kin=None):
        kin = kin or self.kinematics(qpos)
        nv = self.nv
        nt = len(self.tendon_adr)
        L = np.zeros(nt)
        J = np.zeros((nt, nv))
        sx = self.site_xpos(kin)
        for t in range(nt):
            a, n = self.tendon_adr[t], self.tendon_num[t]
            if self.wrap_type[a] == WRAP_JOINT:
                for w in range(a, a + n):     # (wrap_objid is a JOINT id)
                    L[t] += self.wrap_prm[w] * qpos[self.jnt_qposadr[self.wrap_objid[w]]]
                    J[t, self.jnt_dofadr[self.wrap_objid[w]]] = self.wrap_prm[w]
            else:
                for w in range(a, a + n - 1):
                    s0, s1 = self.wrap_objid[w], self.wrap_objid[w + 1]
                    d = sx[s1] - sx[s0]
                    ln = np.linalg.norm(d)
                    L[t] += ln
                    if ln > MJ_MINVAL:
                        u = d / ln
                        j1, _ = self._jac_point(kin, self.site_bodyid[s1], sx[s1])
                        j0, _ = self._jac_point(kin, self.site_bodyid[s0], sx[s0])
                        J[t] += u @ (j1 - j0)
        return L, J

    def _set_const(self):
        """qpos0-dependent constants (MuJoCo's mj_setConst): tendon_length0, *_invweight0, meaninertia."""
        M, kin = self.mass_matrix(self.qpos0)
        nv = M.shape[0]
        Minv = np.linalg.inv(M)
        self.meaninertia = float(np.trace(M) / max(1, nv))
        self.dof_invweight0 = np.diag(Minv).copy()
        for j in np.flatnonzero(np.asarray(self.jnt_type) == JNT_FREE):   # mj_setConst: one value for the 3 translations, one for the 3 rotations
            d = self.jnt_dofadr[j]
            self.dof_invweight0[d:d + 3] = self.dof_invweight0[d:d + 3].mean()
            self.dof_invweight0[d + 3:d + 6] = self.dof_invweight0[d + 3:d + 6].mean()
        nb = len(self.body_parentid)
        self.body_invweight0 = np.zeros((nb, 2))
        for b in range(1, nb):
            if self.body_weldid[b] == 0:
                continue
            com = kin["xpos"][b] + kin["xmat"][b] @ self.body_ipos[b]
            jp, jr = self._jac_point(kin, b, com)
            J = np.vstack([jp, jr])
            nz = np.flatnonzero(np.abs(J).sum(0))
            A = J[:, nz] @ Minv[np.ix_(nz, nz)] @ J[:, nz].T
            self.body_invweight0[b, 0] = (A[0, 0] + A[1, 1] + A[2, 2]) / 3
            self.body_invweight0[b, 1] = (A[3, 3] + A[4, 4] + A[5, 5]) / 3
        L, J = self.tendon_length_jac(self.qpos0, kin)
        self.tendon_length0 = L
        self.tendon_lengthspring = L.copy()  # springlength=-1 -> length at qpos0
        self.tendon_invweight0 = np.array([J[t] @ Minv @ J[t] for t in range(len(L))]).reshape(len(L))

    # ---- sizes ----
    @property
    def nbody(self):
        return len(self.body_parentid)

    @property
    def nv(self):
        return len(self.dof_damping)

    @property
    def nq(self):
        return len(self.qpos0)

    @property
    def njnt(self):
        return len(self.jnt_type)

    # address maps joint -> first position / first dof, dof -> joint.  Identity for models of scalar joints (hinge / slide), whose
    # blobs do not carry them; a free joint (7 positions, 6 dofs) makes the three index spaces differ.
    def _adr(self, name, n):
        v = self.__dict__.get(name)
        return v if v is not None and len(v) == n else np.arange(n, dtype=np.int32)

    jnt_qposadr = property(lambda self: self._adr("_jnt_qposadr", len(self.jnt_type)), lambda self, v: self.__dict__.__setitem__("_jnt_qposadr", np.asarray(v, np.int32)))
    jnt_dofadr = property(lambda self: self._adr("_jnt_dofadr", len(self.jnt_type)), lambda self, v: self.__dict__.__setitem__("_jnt_dofadr", np.asarray(v, np.int32)))
    dof_jntid = property(lambda self: self._adr("_dof_jntid", len(self.dof_damping)), lambda self, v: self.__dict__.__setitem__("_dof_jntid", np.asarray(v, np.int32)))

    @property
    def has_free_joint(self):
        return bool((np.asarray(self.jnt_type) == JNT_FREE).any())

    @property
    def nsite(self):
        return len(self.site_bodyid)

    @property
    def ngeom(self):
        return len(self.geom_type)

    @property
    def ntendon(self):
        return len(self.tendon_adr)

    @property
    def neq(self):
        return len(self.eq_type)

    @property
    def nu(self):
        return len(self.actuator_trnid)

    na = nu

    @property
    def nsensordata(self):
        return 3 * len(self.sensor_type)

    def geom_id2name(self, gid):
        n = self.geom_names[gid]
        return n if n else None

    # ---- serialisation: tagged-array container read by the C ABI and the oracle ----
    _FIELDS_F64 = [
        "body_pos", "body_quat", "body_ipos", "body_imat", "body_mass", "body_invweight0",
        "jnt_pos", "jnt_axis", "jnt_range", "jnt_stiffness", "jnt_margin", "jnt_solref", "jnt_solimp",
        "qpos0", "qpos_spring", "dof_damping", "dof_armature", "dof_invweight0",
        "geom_size", "geom_pos", "geom_quat", "geom_friction", "geom_solref", "geom_solimp", "geom_solmix",
        "geom_margin", "geom_gap", "geom_rbound",
        "site_pos", "site_quat",
        "tendon_stiffness", "tendon_damping", "tendon_lengthspring", "tendon_length0", "tendon_invweight0",
        "wrap_prm", "eq_solref", "eq_solimp", "eq_data",
        "actuator_timeconst", "actuator_gain", "actuator_bias", "actuator_gear",
    ]
    _FIELDS_I32 = [
        "body_parentid", "body_weldid", "body_jntadr", "body_jntnum", "body_geomadr", "body_geomnum",
        "jnt_type", "jnt_bodyid", "jnt_limited", "dof_parentid",
        "geom_type", "geom_bodyid", "geom_contype", "geom_conaffinity", "geom_condim", "geom_priority",
        "site_bodyid", "tendon_adr", "tendon_num", "wrap_type", "wrap_objid",
        "eq_type", "eq_obj1id", "eq_obj2id", "actuator_trnid", "sensor_type", "sensor_objid", "sensor_adr",
    ]

    def to_blob(self) -> bytes:
        recs = []

        def add(name, arr, code):
            raw = np.ascontiguousarray(arr).tobytes()
            pad = (-len(raw)) % 8
            nm = name.encode()
            if len(nm) > 23:
                raise ValueError(name)
            recs.append(struct.pack("<24sIIq", nm, code, 0, np.asarray(arr).size) + raw + b"\0" * pad)

        opt_d = np.array([self.opt_timestep, *self.opt_gravity, self.opt_tolerance, self.opt_impratio,
                          self.meaninertia], dtype=np.float64)
        opt_i = np.array([self.opt_iterations, self.nconmax, self.njmax, int(getattr(self, "opt_implicit_tendon_damping", 0))],
                         dtype=np.int32)
        add("opt_d", opt_d, 1)
        add("opt_i", opt_i, 2)
        for f in self._FIELDS_F64:
            add(f, np.asarray(getattr(self, f), dtype=np.float64), 1)
        for f in self._FIELDS_I32:
            add(f, np.asarray(getattr(self, f), dtype=np.int32), 2)
        if self.has_free_joint:   # only then do joint, position and dof indices differ (blobs of scalar-joint models stay as they were)
            for f in ("jnt_qposadr", "jnt_dofadr", "dof_jntid"):
                add(f, np.asarray(getattr(self, f), dtype=np.int32), 2)
        names = "\n".join(["|".join(self.body_names), "|".join(self.jnt_names), "|".join(self.geom_names),
                           "|".join(self.site_names), "|".join(self.tendon_names), "|".join(self.sensor_names)])
        add("names", np.frombuffer(names.encode(), dtype=np.uint8), 3)
        body = b"".join(recs)
        return struct.pack("<IIIIq", BLOB_MAGIC, BLOB_VERSION, len(recs), 0, len(body) + 24) + body

    @classmethod
    def from_blob(cls, blob: bytes) -> "Model":
        magic, ver, nrec, _, total = struct.unpack_from("<IIIIq", blob, 0)
        if magic != BLOB_MAGIC or ver != BLOB_VERSION or total != len(blob):
            raise ValueError("not a softgrip model blob")
        off = 24
        m = cls()
        for _ in range(nrec):
            nm, code, _, cnt = struct.unpack_from("<24sIIq", blob, off)
            off += 40
            name = nm.rstrip(b"\0").decode()
            dt = {1: np.float64, 2: np.int32, 3: np.uint8}[code]
            nbytes = cnt * np.dtype(dt).itemsize
            arr = np.frombuffer(blob, dtype=dt, count=cnt, offset=off).copy()
            off += nbytes + ((-nbytes) % 8)
            setattr(m, name, arr)
        o = m.opt_d
        m.opt_timestep, m.opt_gravity, m.opt_tolerance, m.opt_impratio, m.meaninertia = o[0], o[1:4], o[4], o[5], o[6]
        m.opt_iterations, m.nconmax, m.njmax = (int(x) for x in m.opt_i[:3])
        m.opt_implicit_tendon_damping = int(m.opt_i[3]) if len(m.opt_i) > 3 else 0
        shp = dict(body_pos=3, body_quat=4, body_ipos=3, body_imat=9, body_invweight0=2, jnt_pos=3, jnt_axis=3,
                   jnt_range=2, jnt_solref=2, jnt_solimp=5, geom_size=3, geom_pos=3, geom_quat=4, geom_friction=3,
                   geom_solref=2, geom_solimp=5, site_pos=3, site_quat=4, eq_solref=2, eq_solimp=5, eq_data=5,
                   actuator_bias=3)
        for k, w in shp.items():
            setattr(m, k, getattr(m, k).reshape(-1, w))
        lines = bytes(m.names).decode().split("\n")
        (m.body_names, m.jnt_names, m.geom_names, m.site_names, m.tendon_names, m.sensor_names) = \
            [ln.split("|") if ln else [] for ln in lines]
        return m

    def dump(self) -> str:
        """Human-readable summary in mj_printModel spirit (for diffing against a real MuJoCo one day)."""
        out = ["nbody %d  nq %d  nv %d  ngeom %d  nsite %d  ntendon %d  neq %d  nu %d  nsensordata %d" % (
            self.nbody, self.nq, self.nv, self.ngeom, len(self.site_bodyid), self.ntendon, self.neq, self.nu,
            self.nsensordata),
            "timestep %.6g  iterations %d  tolerance %.3g  impratio %g  meaninertia %.10g" % (
                self.opt_timestep, self.opt_iterations, self.opt_tolerance, self.opt_impratio, self.meaninertia),
            "total mass %.12g" % self.body_mass.sum()]
        for i in range(self.nbody):
            out.append("BODY %d %s parent %d weld %d mass %.10g pos %s" % (
                i, self.body_names[i], self.body_parentid[i], self.body_weldid[i], self.body_mass[i],
                np.array2string(self.body_pos[i], precision=8)))
        for t in range(self.ntendon):
            out.append("TENDON %d %s length0 %.10g invweight0 %.10g stiffness %g damping %g" % (
                t, self.tendon_names[t], self.tendon_length0[t], self.tendon_invweight0[t],
                self.tendon_stiffness[t], self.tendon_damping[t]))
        return "\n".join(out)


def compile_mjcf(path: str, composite_neighbors: bool = True) -> Model:
    """Counterpart of ``mujoco_py.load_model_from_path`` (reference environment/manenv.py:27).

    ``composite_neighbors`` (default True) adds the neighbour equalities of box / ellipsoid / cylinder composites as MuJoCo's
    composite documentation describes them ("each joint is equality-constrained to remain equal to its
    neighbor joints").  ``False`` compiles the fix-rows-only variant (``models/*_fix.sgmodel``); a MuJoCo capture whose ``neq``
    is 111 instead of 327 would be the evidence to flip the default back (tests/test_mujoco_golden.py)."""
    return _Compiler(path, composite_neighbors).run()


def load_model(path: str, tendon_damper: str = None) -> Model:
    """Load either an MJCF ``.xml`` (compiled here) or a precompiled ``.sgmodel`` blob.

    ``tendon_damper``: ``None`` keeps what the file says (an ``.xml`` compiles to "explicit"); "explicit" is MuJoCo's
    semi-implicit Euler as restated (the damper of a tendon is a passive force evaluated at the old velocity, only joint
    damping enters M + h B: engine_forward.c mj_Euler); "implicit" also integrates the damper of the composite's fixed volume
    tendon implicitly, qacc = (M + h B + h c J'J)^-1 f (deviation D5 of DESIGN.md: the reference's soft ball / cylinder scenes
    start in deep penetration and the explicit damper, c h sum_e 1/(m_e + h d_e) = 215 >> 2, diverges on them)."""
    if path.endswith(".xml"):
        m = compile_mjcf(path)
    else:
        with open(path, "rb") as f:
            m = Model.from_blob(f.read())
    if tendon_damper is not None:
        if tendon_damper not in ("explicit", "implicit"):
            raise ValueError("tendon_damper must be 'explicit' or 'implicit'")
        m.opt_implicit_tendon_damping = int(tendon_damper == "implicit")
    return m
