"""Parameter table of the 15-body / 28-dof tracking humanoid and an MJCF writer for it.

The tracker's character is configured by ``char_file`` in the env YAML (reference default:
``data/assets/humanoid.xml``).  A user of the reference points ``char_file`` at their own copy; for
tests, smoke and bench this module regenerates an equivalent MJCF from the table below
(``write_mjcf``), so the repo carries the character as data rather than as a copied asset file.
tests/test_host_logic.py checks the parsed tree against the reference's own parse (golden g2_char).

Units: metres, degrees (MJCF convention), kg/m^3, N*m/rad, N*m*s/rad.
"""
import os

# joint spec: (axis letter -> (lo, hi)), stiffness, damping, armature, gears per axis
#   a body with one axis is a hinge, with three axes (x, y, z) a spherical joint.
# geom spec: ("sphere", pos, radius, density) | ("capsule", from, to, radius, density) | ("box", pos, half, density)
BODIES = [
    dict(name="pelvis", parent=None, pos=(0, 0, 0), joint=None,
         geoms=[("sphere", (0, 0, 0.07), 0.09, 2226), ("sphere", (0, 0, 0.205), 0.07, 2226)]),
    dict(name="torso", parent="pelvis", pos=(0, 0, 0.236151),
         joint=dict(base="abdomen", axes=dict(x=(-60, 60), y=(-60, 90), z=(-70, 70)), kp=1000, kd=100, arm=0.02,
                    gear=dict(x=200, y=200, z=200)),
         geoms=[("sphere", (0, 0, 0.12), 0.11, 1794),
                ("capsule", (-0.0060125, -0.0457775, 0.2287955), (-0.016835, -0.128177, 0.2376182), 0.045, 1100),
                ("capsule", (-0.0060125, 0.0457775, 0.2287955), (-0.016835, 0.128177, 0.2376182), 0.045, 1100)]),
    dict(name="head", parent="torso", pos=(0, 0, 0.223894),
         joint=dict(base="neck", axes=dict(x=(-50, 50), y=(-40, 60), z=(-45, 45)), kp=100, kd=10, arm=0.01,
                    gear=dict(x=50, y=50, z=50)),
         geoms=[("sphere", (0, 0, 0.175), 0.095, 1081)]),
    dict(name="right_upper_arm", parent="torso", pos=(-0.02405, -0.18311, 0.24350),
         joint=dict(base="right_shoulder", axes=dict(x=(-110, 140), y=(-90, 90), z=(-60, 160)), kp=400, kd=40, arm=0.02,
                    gear=dict(x=100, y=100, z=100)),
         geoms=[("capsule", (0, -0.03, 0), (0, -0.23, 0), 0.045, 982)]),
    dict(name="right_lower_arm", parent="right_upper_arm", pos=(0, -0.274788, 0),
         joint=dict(base="right_elbow", axes=dict(z=(0, 160)), kp=300, kd=30, arm=0.01, gear=dict(z=70)),
         geoms=[("capsule", (0, -0.035, 0), (0, -0.1875, 0), 0.04, 1056)]),
    dict(name="right_hand", parent="right_lower_arm", pos=(0, -0.258947, 0), joint=None,
         geoms=[("sphere", (0, 0, 0), 0.04, 1865)]),
    dict(name="left_upper_arm", parent="torso", pos=(-0.02405, 0.18311, 0.24350),
         joint=dict(base="left_shoulder", axes=dict(x=(-140, 110), y=(-90, 90), z=(-160, 60)), kp=400, kd=40, arm=0.02,
                    gear=dict(x=100, y=100, z=100)),
         geoms=[("capsule", (0, 0.03, 0), (0, 0.23, 0), 0.045, 982)]),
    dict(name="left_lower_arm", parent="left_upper_arm", pos=(0, 0.274788, 0),
         joint=dict(base="left_elbow", axes=dict(z=(-160, 0)), kp=300, kd=30, arm=0.01, gear=dict(z=70)),
         geoms=[("capsule", (0, 0.035, 0), (0, 0.1875, 0), 0.04, 1056)]),
    dict(name="left_hand", parent="left_lower_arm", pos=(0, 0.258947, 0), joint=None,
         geoms=[("sphere", (0, 0, 0), 0.04, 1865)]),
    dict(name="right_thigh", parent="pelvis", pos=(0, -0.084887, 0),
         joint=dict(base="right_hip", axes=dict(x=(-60, 30), y=(-140, 60), z=(-60, 60)), kp=500, kd=50, arm=0.02,
                    gear=dict(x=200, y=200, z=200)),
         geoms=[("capsule", (0, 0, -0.04), (0, 0, -0.36), 0.055, 1269)]),
    dict(name="right_shin", parent="right_thigh", pos=(0, 0, -0.421546),
         joint=dict(base="right_knee", axes=dict(y=(0, 160)), kp=500, kd=50, arm=0.02, gear=dict(y=150)),
         geoms=[("capsule", (0, 0, -0.03), (0, 0, -0.355), 0.05, 1014)]),
    dict(name="right_foot", parent="right_shin", pos=(0, 0, -0.409870),
         joint=dict(base="right_ankle", axes=dict(x=(-30, 30), y=(-55, 55), z=(-40, 40)), kp=400, kd=40, arm=0.01,
                    gear=dict(x=90, y=90, z=90)),
         geoms=[("box", (0.045, 0, -0.0225), (0.0885, 0.045, 0.0275), 1141)]),
    dict(name="left_thigh", parent="pelvis", pos=(0, 0.084887, 0),
         joint=dict(base="left_hip", axes=dict(x=(-30, 60), y=(-140, 60), z=(-60, 60)), kp=500, kd=50, arm=0.02,
                    gear=dict(x=200, y=200, z=200)),
         geoms=[("capsule", (0, 0, -0.04), (0, 0, -0.36), 0.055, 1269)]),
    dict(name="left_shin", parent="left_thigh", pos=(0, 0, -0.421546),
         joint=dict(base="left_knee", axes=dict(y=(0, 160)), kp=500, kd=50, arm=0.02, gear=dict(y=150)),
         geoms=[("capsule", (0, 0, -0.03), (0, 0, -0.355), 0.05, 1014)]),
    dict(name="left_foot", parent="left_shin", pos=(0, 0, -0.409870),
         joint=dict(base="left_ankle", axes=dict(x=(-30, 30), y=(-55, 55), z=(-40, 40)), kp=400, kd=40, arm=0.01,
                    gear=dict(x=90, y=90, z=90)),
         geoms=[("box", (0.045, 0, -0.0225), (0.0885, 0.045, 0.0275), 1141)]),
]

_AXIS = dict(x="1 0 0", y="0 1 0", z="0 0 1")


def _num(x):
    """shortest decimal that reads back as the same double (no 6-digit rounding of the geometry)"""
    r = repr(float(x))
    return r[:-2] if r.endswith(".0") else r


def _fmt(v):
    return " ".join(_num(x) for x in v)


def _joint_names(j):
    axes = j["axes"]
    if len(axes) == 1:
        return {a: j["base"] for a in axes}
    return {a: "{}_{}".format(j["base"], a) for a in axes}


def _body_xml(spec, children, indent):
    pad = "  " * indent
    lines = []
    if spec["parent"] is None:
        lines.append('{}<body name="{}" pos="{}" childclass="body">'.format(pad, spec["name"], _fmt(spec["pos"])))
        lines.append('{}  <freejoint name="root"/>'.format(pad))
    else:
        lines.append('{}<body name="{}" pos="{}">'.format(pad, spec["name"], _fmt(spec["pos"])))
    j = spec["joint"]
    if j is not None:
        names = _joint_names(j)
        for a, (lo, hi) in j["axes"].items():
            lines.append('{}  <joint name="{}" type="hinge" axis="{}" range="{} {}" stiffness="{}" damping="{}" armature="{}"/>'.format(
                pad, names[a], _AXIS[a], _num(lo), _num(hi), _num(j["kp"]), _num(j["kd"]), _num(j["arm"])))
    for gi, g in enumerate(spec["geoms"]):
        gname = spec["name"] if gi == 0 else "{}_g{}".format(spec["name"], gi)
        if g[0] == "sphere":
            lines.append('{}  <geom name="{}" type="sphere" pos="{}" size="{}" density="{}"/>'.format(pad, gname, _fmt(g[1]), _num(g[2]), _num(g[3])))
        elif g[0] == "capsule":
            lines.append('{}  <geom name="{}" type="capsule" fromto="{} {}" size="{}" density="{}"/>'.format(
                pad, gname, _fmt(g[1]), _fmt(g[2]), _num(g[3]), _num(g[4])))
        elif g[0] == "box":
            lines.append('{}  <geom name="{}" type="box" pos="{}" size="{}" density="{}"/>'.format(pad, gname, _fmt(g[1]), _fmt(g[2]), _num(g[3])))
    for c in children.get(spec["name"], []):
        lines.extend(_body_xml(c, children, indent + 1))
    lines.append("{}</body>".format(pad))
    return lines


def mjcf_text():
    children = {}
    for b in BODIES:
        if b["parent"] is not None:
            children.setdefault(b["parent"], []).append(b)
    out = ['<mujoco model="humanoid">', '  <default>', '    <motor ctrlrange="-1 1" ctrllimited="true"/>',
           '    <default class="body">', '      <geom type="capsule" condim="1" friction="1.0 0.05 0.05"/>',
           '      <joint type="hinge" damping="0.1" stiffness="5" armature=".007" limited="true"/>',
           '    </default>', '  </default>', '  <worldbody>']
    out.extend(_body_xml(BODIES[0], children, 2))
    out.append('  </worldbody>')
    out.append('  <actuator>')
    for b in BODIES:
        j = b["joint"]
        if j is None:
            continue
        names = _joint_names(j)
        for a in j["axes"]:
            out.append('    <motor name="{0}" gear="{1:g}" joint="{0}"/>'.format(names[a], j["gear"][a]))
    out.append('  </actuator>')
    out.append('</mujoco>')
    return "\n".join(out) + "\n"


def write_mjcf(path=None):
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "humanoid.xml")
    text = mjcf_text()
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
    return path
