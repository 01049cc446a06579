"""Seeded generator of the synthetic large scene as a legal `.test` file.

The file goes through the ordinary front end (rt_parse_input_file, or the reference's
parse_input_file: same format, Source/input_parsing.cpp:96-603), so nothing here bypasses the
parser.  Default = the benchmark configuration: 1,000,000 triangles + 10,000 ellipsoids,
3840x2160, 8 point lights, maxdepth 2.  Smaller instances are used by the tests.

Numbers are printed with at most 6 decimals and no exponent: the format's number syntax has no
exponent notation (input_parsing.cpp:25-52).
"""
import math
import os
import tempfile

import numpy as np


def _fmt(a):
    """Rows of floats -> list of 'x y z' strings with 6 decimals (trailing zeros trimmed)."""
    return [" ".join(("%.6f" % v).rstrip("0").rstrip(".") or "0" for v in row) for row in a]


def write_scene(path, n_triangles=1_000_000, n_ellipsoids=10_000, n_lights=8, width=3840, height=2160,
                maxdepth=2, seed=1, extent=10.0, edge=0.15, materials_every=1000, ell_scale=(0.05, 0.3),
                n_spheres=0, sphere_radius=(0.05, 0.3), directional=None, attenuation=(1.0, 0.0, 0.0), eye=None, fov=45.0):
    """Writes the scene; returns the path.  Deterministic in its arguments.
    directional: None, or the direction (x, y, z) of a directional light of colour .1 (the parser normalises it);
    attenuation: the three coefficients of the point lights' attenuation; eye: the camera position (default: outside the cloud, at
    (0, 0, 4 extent)), looking at the centre; fov: its vertical field of view in degrees."""
    rng_t = np.random.Generator(np.random.PCG64(seed))
    rng_e = np.random.Generator(np.random.PCG64(seed + 1))
    rng_m = np.random.Generator(np.random.PCG64(seed + 2))
    rng_s = np.random.Generator(np.random.PCG64(seed + 3))
    lines = []
    add = lines.append
    add("# synthetic scene: %d triangles, %d ellipsoids, %d point lights, seed %d" % (n_triangles, n_ellipsoids, n_lights, seed))
    add("size %d %d" % (width, height))
    add("maxdepth %d" % maxdepth)
    add("output synthetic.png")
    if eye is None and fov == 45.0:
        add("camera 0 0 %s 0 0 0 0 1 0 45" % ("%.6f" % (4.0 * extent)).rstrip("0").rstrip("."))
    else:
        add("camera %s 0 0 0 0 1 0 %s" % (_fmt([eye if eye is not None else (0.0, 0.0, 4.0 * extent)])[0], _fmt([[fov]])[0]))
    add("ambient .05 .05 .05")
    add("attenuation " + _fmt([attenuation])[0])
    if directional is not None:
        add("directional " + _fmt([directional])[0] + " .1 .1 .1")
    for i in range(n_lights):
        ang = 2.0 * math.pi * i / max(n_lights, 1)
        add("point %.6f %.6f %.6f .15 .15 .15" % (2.5 * extent * math.cos(ang), 0.8 * extent, 2.5 * extent * math.sin(ang)))
    add("shininess 20")
    add("maxverts %d" % (3 * n_triangles + 1))

    def material():
        d = rng_m.uniform(0.1, 0.5, 3)
        s = rng_m.uniform(0.1, 0.5, 3)
        return ["diffuse %.4f %.4f %.4f" % tuple(d), "specular %.4f %.4f %.4f" % tuple(s)]

    # triangles: centroid uniform in the cube, two edge vectors uniform in [-edge, edge]^3
    chunk = 100_000
    done = 0
    while done < n_triangles:
        n = min(chunk, n_triangles - done)
        c = rng_t.uniform(-extent, extent, (n, 3))
        e1 = rng_t.uniform(-edge, edge, (n, 3))
        e2 = rng_t.uniform(-edge, edge, (n, 3))
        v0 = c - (e1 + e2) / 3.0
        s0, s1, s2 = _fmt(v0), _fmt(v0 + e1), _fmt(v0 + e2)
        for i in range(n):
            g = done + i
            if g % materials_every == 0:
                lines.extend(material())
            lines.append("vertex " + s0[i])
            lines.append("vertex " + s1[i])
            lines.append("vertex " + s2[i])
            lines.append("tri %d %d %d" % (3 * g, 3 * g + 1, 3 * g + 2))
        done += n

    # ellipsoids: unit sphere under translate * rotate * scale with three clearly different scales
    for g in range(n_ellipsoids):
        if g % materials_every == 0:
            lines.extend(material())
        t = rng_e.uniform(-extent, extent, 3)
        axis = rng_e.normal(size=3)
        axis /= np.linalg.norm(axis)
        angle = rng_e.uniform(0.0, 360.0)
        while True:
            sc = rng_e.uniform(ell_scale[0], ell_scale[1], 3)
            if min(abs(sc[0] - sc[1]), abs(sc[1] - sc[2]), abs(sc[0] - sc[2])) > 0.02 * (ell_scale[1] - ell_scale[0]):
                break
        lines.append("pushTransform")
        lines.append("translate %.6f %.6f %.6f" % tuple(t))
        lines.append("rotate %.6f %.6f %.6f %.4f" % (axis[0], axis[1], axis[2], angle))
        lines.append("scale %.6f %.6f %.6f" % tuple(sc))
        lines.append("sphere 0 0 0 1")
        lines.append("popTransform")

    # true spheres (no transform in force): centre uniform in the cube, radius uniform in sphere_radius
    for g in range(n_spheres):
        if g % materials_every == 0:
            lines.extend(material())
        c = rng_s.uniform(-extent, extent, 3)
        lines.append("sphere %.6f %.6f %.6f %.6f" % (c[0], c[1], c[2], rng_s.uniform(sphere_radius[0], sphere_radius[1])))

    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        f.write("\n".join(lines))
        f.write("\n")
    os.replace(tmp, path)
    return path


def ensure_scene_file(directory=None, **kwargs):
    """Path of the scene for these parameters, generated on first use (about a minute for the full size)."""
    directory = directory or os.environ.get("RT_SYNTH_DIR") or tempfile.gettempdir()
    tag = "_".join("%s%s" % (k[:3], str(kwargs[k]).replace(" ", "").replace("(", "").replace(")", "").replace(",", "-")) for k in sorted(kwargs)) or "default"
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "rt_synthetic_%s.test" % tag)
    if not os.path.exists(path):
        write_scene(path, **kwargs)
    return path
