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
