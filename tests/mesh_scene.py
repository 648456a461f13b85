"""Synthetic triangle-mesh scene for the BVH tests and bench.py --workload mesh."""
import os
import pathlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_mesh_scene(tmp_path, n_lat=48, n_lon=96, poles=False, width=160, height=120):
    """A tessellated glass blob over a displaced ground grid: ~11k triangles with shared edges,
    grazing silhouettes and overlapping meshes (equal-distance ties at shared edges are decided by
    scene order, which the BVH must reproduce)."""
    import math
    lines = []
    nv = 0

    def grid(fn, nu, nv_, normal_fn):
        nonlocal nv
        base = nv
        for i in range(nu + 1):
            for j in range(nv_ + 1):
                p = fn(i / nu, j / nv_)
                nrm = normal_fn(i / nu, j / nv_)
                lines.append("v %.6f %.6f %.6f" % p)
                lines.append("vn %.6f %.6f %.6f" % nrm)
                lines.append("vt %.6f %.6f" % (i / nu * 4, j / nv_ * 4))
                nv += 1
        for i in range(nu):
            for j in range(nv_):
                a = base + i * (nv_ + 1) + j + 1
                b, c, d = a + 1, a + nv_ + 1, a + nv_ + 2
                lines.append("f %d/%d/%d %d/%d/%d %d/%d/%d" % (a, a, a, c, c, c, b, b, b))
                lines.append("f %d/%d/%d %d/%d/%d %d/%d/%d" % (b, b, b, c, c, c, d, d, d))

    cap = 0.0 if poles else 0.02   # poles=True closes the blob with zero-area triangles at the poles

    def blob(u, v):
        th, ph = math.pi * (cap + (1 - 2 * cap) * u), 2 * math.pi * v
        r = 1.0 + 0.15 * math.sin(5 * th) * math.cos(3 * ph)
        return (r * math.sin(th) * math.cos(ph), r * math.cos(th), r * math.sin(th) * math.sin(ph) - 5.0)

    def blob_n(u, v):
        th, ph = math.pi * (cap + (1 - 2 * cap) * u), 2 * math.pi * v
        return (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph))
    grid(blob, n_lat, n_lon, blob_n)
    (tmp_path / "blob.obj").write_text("\n".join(lines) + "\n")
    lines.clear()
    nv = 0
    grid(lambda u, v: (-6 + 12 * u, -1.6 + 0.2 * math.sin(9 * u) * math.cos(7 * v), -11 + 12 * v), 32, 32,
         lambda u, v: (0.0, 1.0, 0.0))
    (tmp_path / "ground.obj").write_text("\n".join(lines) + "\n")
    xml = tmp_path / "mesh.xml"
    xml.write_text('''<scene output_file="mesh.png"><background_color r="0.1" g="0.1" b="0.2"/>
  <camera><position x="0" y="0.5" z="0"/><lookat x="0" y="0" z="-5"/><up x="0" y="1" z="0"/><horizontal_fov angle="40"/>
    <resolution horizontal="%d" vertical="%d"/><max_bounces n="4"/></camera>
  <lights><ambient_light><color r="1" g="1" b="1"/></ambient_light>
    <point_light><color r="0.7" g="0.7" b="0.7"/><position x="3" y="6" z="-1"/></point_light>
    <parallel_light><color r="0.3" g="0.3" b="0.3"/><direction x="-1" y="-1" z="-1"/></parallel_light></lights>
  <surfaces>
    <mesh name="ground.obj"><material_textured><texture name="%s"/><phong ka="0.3" kd="0.9" ks="1.0" exponent="50"/>
      <reflectance r="0"/><transmittance t="0"/><refraction iof="0"/></material_textured></mesh>
    <mesh name="blob.obj"><material_solid><color r="0.2" g="0.4" b="0.3"/><phong ka="0.2" kd="0.5" ks="1.0" exponent="100"/>
      <reflectance r="0.5"/><transmittance t="0.5"/><refraction iof="1.4"/></material_solid></mesh>
    <sphere radius="0.5"><position x="1.8" y="-0.5" z="-4"/><material_solid><color r="0.8" g="0.2" b="0.2"/>
      <phong ka="0.3" kd="0.9" ks="1.0" exponent="200"/><reflectance r="0.4"/><transmittance t="0"/><refraction iof="2.3"/></material_solid></sphere>
  </surfaces></scene>''' % (width, height, os.path.join(ROOT, "scenes", "examples", "mramor6x6.png")))
    return str(xml)


def write_culling_scene(tmp_path, width=320, height=200):
    """Spheres and small meshes far from the world origin seen from a distant camera: the case where
    the reference's object-space discriminant is dominated by rounding error, which the bounding-sphere
    and chunk culling must never turn into a different answer (slack terms of DevSphereBound /
    DevTriChunk).  Squashed, rotated ellipsoids; one sphere encloses the camera; a light inside a sphere."""
    box = os.path.join(ROOT, "scenes", "examples", "box.obj")
    plane = os.path.join(ROOT, "scenes", "examples", "plane_small.obj")
    spheres = []
    for i in range(9):
        x, y = (i % 3 - 1) * 7.0, (i // 3 - 1) * 5.0
        spheres.append('''<sphere radius="%.2f"><position x="0" y="0" z="0"/><material_solid><color r="%.2f" g="%.2f" b="0.3"/>
      <phong ka="0.3" kd="0.8" ks="1.0" exponent="%d"/><reflectance r="%s"/><transmittance t="%s"/><refraction iof="%s"/></material_solid>
      <transform><translate x="%.1f" y="%.1f" z="%.1f"/><rotateZ theta="%d"/><scale x="%.2f" y="%.2f" z="%.2f"/><rotateY theta="%d"/></transform></sphere>''' % (
            0.5 + 0.2 * i, 0.2 + 0.08 * i, 0.9 - 0.08 * i, 20 + 30 * i, "0.5" if i % 3 == 0 else "0", "0.7" if i % 3 == 1 else "0",
            "1.5" if i % 3 else "2.0", 4000.0 + x, 3000.0 + y, -5000.0 - 3 * i, 17 * i, 1.0 + 0.6 * i, 1.0 / (1.0 + 0.3 * i), 0.4 + 0.2 * i, 23 * i))
    meshes = []
    for i in range(6):
        meshes.append('''<mesh name="%s"><material_solid><color r="0.6" g="%.2f" b="0.2"/><phong ka="0.3" kd="0.9" ks="1.0" exponent="60"/>
      <reflectance r="0"/><transmittance t="0"/><refraction iof="0"/></material_solid>
      <transform><translate x="%.1f" y="%.1f" z="%.1f"/><rotateY theta="%d"/><rotateX theta="%d"/><scale x="%.2f" y="1.0" z="%.2f"/></transform></mesh>''' % (
            box, 0.1 + 0.15 * i, 4000.0 - 9.0 + 3.5 * i, 3000.0 - 6.0 + (i % 2) * 11.0, -5004.0 - i, 31 * i, 11 * i, 0.5 + 0.3 * i, 1.5 - 0.2 * i))
    xml = tmp_path / "culling.xml"
    xml.write_text('''<scene output_file="culling.png"><background_color r="0.05" g="0.05" b="0.1"/>
  <camera><position x="4000" y="3001" z="-4940"/><lookat x="4000" y="3000" z="-5004"/><up x="0" y="1" z="0"/><horizontal_fov angle="12"/>
    <resolution horizontal="%d" vertical="%d"/><max_bounces n="4"/></camera>
  <lights><ambient_light><color r="1" g="1" b="1"/></ambient_light>
    <point_light><color r="0.8" g="0.8" b="0.8"/><position x="4010" y="3015" z="-4990"/></point_light>
    <point_light><color r="0.4" g="0.4" b="0.5"/><position x="4000" y="3000" z="-5000.2"/></point_light>
    <parallel_light><color r="0.3" g="0.3" b="0.3"/><direction x="-1" y="-1" z="-1"/></parallel_light></lights>
  <surfaces>
    <mesh name="%s"><material_solid><color r="0.5" g="0.5" b="0.5"/><phong ka="0.3" kd="0.9" ks="0.5" exponent="30"/>
      <reflectance r="0"/><transmittance t="0"/><refraction iof="0"/></material_solid>
      <transform><translate x="4000" y="2990" z="-5000"/><rotateX theta="-90"/><scale x="40" y="40" z="1"/></transform></mesh>
    %s
    %s
    <sphere radius="200"><position x="0" y="0" z="0"/><material_solid><color r="0.2" g="0.2" b="0.4"/>
      <phong ka="0.5" kd="0.5" ks="0.0" exponent="1"/><reflectance r="0"/><transmittance t="0"/><refraction iof="0"/></material_solid>
      <transform><translate x="4000" y="3000" z="-5000"/></transform></sphere>
  </surfaces></scene>''' % (width, height, plane, "\n    ".join(meshes), "\n    ".join(spheres)))
    return str(xml)


def write_two_julia_scene(tmp_path, width=120, height=90):
    """Two Julia sets (one behind the other, one scaled and rotated) over a mirror floor with two lights:
    queries that march one set and then the next, shadow rays with a finite maximum, ties by scene order."""
    plane = os.path.join(ROOT, "scenes", "examples", "plane_small.obj")
    xml = tmp_path / "two_julia.xml"
    xml.write_text('''<scene output_file="two_julia.png"><background_color r="0.2" g="0.3" b="0.5"/>
  <camera><position x="0" y="0.5" z="3.2"/><lookat x="0" y="0" z="0"/><up x="0" y="1" z="0"/><horizontal_fov angle="35"/>
    <resolution horizontal="%d" vertical="%d"/><max_bounces n="3"/></camera>
  <lights><ambient_light><color r="1" g="1" b="1"/></ambient_light>
    <point_light><color r="0.8" g="0.8" b="0.8"/><position x="3" y="4" z="3"/></point_light>
    <point_light><color r="0.5" g="0.4" b="0.3"/><position x="-2" y="1" z="1.5"/></point_light></lights>
  <surfaces>
    <julia scale="1.1" cr="-0.2" ca="0.6" cb="0.2" cc="0.2" cutplane="0.0"><position x="-0.6" y="0" z="0"/>
      <material_solid><color r="0.8" g="0.5" b="0.2"/><phong ka="0.3" kd="0.8" ks="1.0" exponent="40"/>
      <reflectance r="0.4"/><transmittance t="0"/><refraction iof="1.6" ec="0.5"/></material_solid></julia>
    <mesh name="%s"><material_solid><color r="0.4" g="0.4" b="0.4"/><phong ka="0.3" kd="0.7" ks="1.0" exponent="100"/>
      <reflectance r="0.6"/><transmittance t="0"/><refraction iof="1.5" ec="1.0"/></material_solid>
      <transform><translate x="0" y="-1.3" z="0"/><rotateX theta="-90"/><scale x="6" y="6" z="1"/></transform></mesh>
    <julia scale="0.9" cr="-0.291" ca="-0.399" cb="0.339" cc="0.437" cutplane="0.1"><position x="0" y="0" z="0"/>
      <material_solid><color r="0.3" g="0.7" b="0.5"/><phong ka="0.3" kd="0.8" ks="1.0" exponent="60"/>
      <reflectance r="0"/><transmittance t="0"/><refraction iof="0"/></material_solid>
      <transform><translate x="0.9" y="0.1" z="-0.8"/><rotateY theta="40"/><scale x="1.2" y="0.9" z="1.0"/></transform></julia>
  </surfaces></scene>''' % (width, height, plane))
    return str(xml)


def write_random_scene(tmp_path, seed):
    """A seeded random scene over the whole grammar of the path: spheres under random transform chains,
    meshes, solid and textured materials with every combination of the reflect / transmit / complex index /
    dispersion switches, 0-3 lights of both kinds, ragged image sizes, supersampling, recursion depths
    0-5, now and then a photon pass or a Julia set.  For the GPU-against-oracle sweep."""
    import random
    rnd = random.Random(seed)
    ex = os.path.join(ROOT, "scenes", "examples")
    u = rnd.uniform
    # one scene in five sits far from the world origin, where the reference's own float arithmetic gets coarse and
    # every cull has to stay on the safe side of it
    far = rnd.choice([60.0, 700.0, 5000.0]) if seed % 5 == 2 else 0.0
    ox, oy, oz = far * 0.8, far * 0.6, -far

    def material():
        textured = rnd.random() < 0.3
        refl = rnd.choice(["0", "0", "0.5", "1.0"])
        trans = rnd.choice(["0", "0", "0.7"])
        iof = rnd.choice(["0", "1.3", "1.5168", "2.3"]) if (refl != "0" or trans != "0") else rnd.choice(["0", "1.5"])
        ec = rnd.choice(["", "", ' ec="1.0"', ' ec="9.7525e-9"'])
        disp = ' disp="-0.2"' if rnd.random() < 0.12 and iof != "0" else ""
        phong = '<phong ka="%.2f" kd="%.2f" ks="%.2f" exponent="%d"/>' % (u(0, 0.5), u(0.2, 1.0), u(0, 1.0), rnd.choice([1, 10, 50, 200, 1000]))
        tail = '%s<reflectance r="%s"/><transmittance t="%s"/><refraction iof="%s"%s%s/>' % (phong, refl, trans, iof, ec, disp)
        if textured:
            return '<material_textured><texture name="%s"/>%s</material_textured>' % (os.path.join(ex, rnd.choice(["mramor6x6.png", "rainbow.png"])), tail)
        return '<material_solid><color r="%.2f" g="%.2f" b="%.2f"/>%s</material_solid>' % (u(0, 1), u(0, 1), u(0, 1), tail)

    def transform(spread):
        ops = ['<translate x="%.3f" y="%.3f" z="%.3f"/>' % (ox + u(-spread, spread), oy + u(-1, 1.5), oz + u(-spread, spread))]
        for _ in range(rnd.randrange(0, 4)):
            k = rnd.randrange(0, 4)
            if k == 3:
                flip = (lambda: rnd.choice([-1.0, 1.0])) if seed % 11 == 4 else (lambda: 1.0)   # mirrored objects now and then
                ops.append('<scale x="%.3f" y="%.3f" z="%.3f"/>' % (flip() * u(0.4, 1.8), flip() * u(0.4, 1.8), flip() * u(0.4, 1.8)))
            else:
                ops.append('<rotate%s theta="%.1f"/>' % ("XYZ"[k], u(-180, 180)))
        return "<transform>%s</transform>" % "".join(ops)

    with_photons = seed % 4 == 1
    diffuse = ('<material_solid><color r="0.6" g="0.6" b="0.6"/><phong ka="0.3" kd="0.8" ks="0.3" exponent="20"/><reflectance r="0"/>'
               '<transmittance t="0"/><refraction iof="0"/></material_solid>')
    glass = ('<material_solid><color r="0.2" g="0.3" b="0.5"/><phong ka="0.1" kd="0.2" ks="1.0" exponent="200"/><reflectance r="0.8"/>'
             '<transmittance t="0.8"/><refraction iof="1.5168" ec="9.7525e-9"%s/></material_solid>' % (' disp="-0.2"' if seed % 8 == 1 else ""))
    surfaces = ['<mesh name="%s">%s<transform><translate x="%.3f" y="%.3f" z="%.3f"/><rotateX theta="-90"/><scale x="8" y="8" z="1"/></transform></mesh>'
                % (os.path.join(ex, "plane_small.obj"), diffuse if with_photons else material(), ox, oy - 1.5, oz)]
    if with_photons:   # something that focuses light onto the diffuse floor
        surfaces.append('<sphere radius="0.8"><position x="0" y="0" z="0"/>%s<transform><translate x="%.2f" y="%.2f" z="%.2f"/><scale x="1.0" y="%.2f" z="1.0"/></transform></sphere>'
                        % (glass, ox + u(-1, 1), oy + 0.2, oz + u(-1, 1), u(0.6, 1.4)))
    for _ in range(rnd.randrange(1, 6)):
        surfaces.append('<sphere radius="%.3f"><position x="%.2f" y="%.2f" z="%.2f"/>%s%s</sphere>'
                        % (u(0.3, 1.0), u(-0.5, 0.5), u(-0.5, 0.5), u(-0.5, 0.5), material(), transform(2.5)))
    for _ in range(rnd.randrange(0, 3)):
        surfaces.append('<mesh name="%s">%s%s</mesh>' % (os.path.join(ex, rnd.choice(["box.obj", "cone.obj"])), material(), transform(2.5)))
    julia = seed % 6 == 5
    if julia:
        surfaces.append('<julia scale="1.2" cr="%.3f" ca="%.3f" cb="%.3f" cc="%.3f" cutplane="%.2f"><position x="%.3f" y="%.3f" z="%.3f"/>%s</julia>'
                        % (u(-0.4, 0.1), u(-0.5, 0.6), u(-0.3, 0.4), u(-0.2, 0.5), u(-0.2, 0.2), ox, oy + 0.3, oz, material()))
    if seed % 3 == 0:
        # the same surface twice with another material: every hit on it is an exact tie, decided by scene order
        # (an object.intersect(ray, max) at t == max is a hit, so the later object wins: src/raytracer.cpp:128-141)
        import re
        twin = rnd.choice(surfaces)
        other = material()
        surfaces.append(re.sub(r"<material_(solid|textured)>.*?</material_(solid|textured)>", lambda _m: other, twin, count=1, flags=re.S))
    rnd.shuffle(surfaces)
    lights = ['<ambient_light><color r="%.2f" g="%.2f" b="%.2f"/></ambient_light>' % (u(0.5, 1), u(0.5, 1), u(0.5, 1))]
    if with_photons:
        lights.append('<point_light><color r="0.8" g="0.8" b="0.8"/><position x="%.2f" y="%.2f" z="%.2f"/></point_light>' % (ox + u(-1, 1), oy + 4, oz + u(-1, 1)))
    for _ in range(rnd.randrange(0, 4)):
        if rnd.random() < 0.6:
            lights.append('<point_light><color r="%.2f" g="%.2f" b="%.2f"/><position x="%.2f" y="%.2f" z="%.2f"/></point_light>'
                          % (u(0.2, 0.9), u(0.2, 0.9), u(0.2, 0.9), ox + u(-4, 4), oy + u(1, 5), oz + u(-2, 4)))
        else:
            lights.append('<parallel_light><color r="%.2f" g="%.2f" b="%.2f"/><direction x="%.2f" y="%.2f" z="%.2f"/></parallel_light>'
                          % (u(0.2, 0.6), u(0.2, 0.6), u(0.2, 0.6), u(-1, 1), u(-1, -0.2), u(-1, 1)))
    w, h = (rnd.choice([48, 57]), rnd.choice([32, 41])) if julia else (rnd.choice([64, 83, 96]), rnd.choice([48, 57, 71]))
    caustic = '<caustic steps="%d" texture_size="%d" factor="0.3"/>' % (rnd.choice([90, 150]), rnd.choice([8, 24])) if with_photons else ""
    # depth of field: the lens jitter is a counter-based hash here and in the oracle (the reference's is not reproducible)
    dof = '<dof x="%.2f" y="%.2f" z="%.2f" lenssize="%.2f"/>' % (ox + u(-0.5, 0.5), oy + u(-0.5, 0.5), oz + u(-1, 1), u(0.05, 0.3)) if seed % 7 == 3 else ""
    xml = tmp_path / ("random_%d.xml" % seed)
    xml.write_text('''<scene output_file="random_%d.png"><background_color r="%.2f" g="%.2f" b="%.2f"/>%s
  <camera><position x="%.2f" y="%.2f" z="%.2f"/><lookat x="%.2f" y="%.2f" z="%.2f"/><up x="0" y="1" z="0"/><horizontal_fov angle="%d"/>
    <resolution horizontal="%d" vertical="%d"/><max_bounces n="%d"/><supersampling subpixels_peraxis="%d"/>%s</camera>
  <lights>%s</lights>
  <surfaces>%s</surfaces></scene>''' % (seed, u(0, 0.5), u(0, 0.5), u(0, 0.5), caustic, ox + u(-2, 2), oy + u(0.2, 2.5), oz + u(4, 6), ox + u(-0.5, 0.5), oy + u(-0.5, 0.5), oz,
                                         rnd.choice([30, 45, 60]), w, h, rnd.randrange(2, 6) if with_photons else rnd.randrange(0, 6),
                                         2 if dof else rnd.choice([1, 1, 2]), dof, "".join(lights), "\n    ".join(surfaces)))
    return str(xml)
