"""The C-ABI library loads and exports every symbol include/rtc.h declares (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

from scene_feed import rtc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rtc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(rtc_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_boundary():
    names = _declared_functions()
    for required in ("rtc_scene_create", "rtc_scene_commit", "rtc_render", "rtc_render_device", "rtc_scene_add_shape",
                     "rtc_scene_begin_group", "rtc_scene_add_triangles", "rtc_matrix_inverse", "rtc_view_transform"):
        assert required in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(rtc._native.LIB_PATH)
    missing = [n for n in _declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_covers_the_header():
    bound = {name for name, _, _ in rtc._native.SYMBOLS}
    assert set(_declared_functions()) <= bound, sorted(set(_declared_functions()) - bound)
    rtc._native.lib()  # resolves every entry of the table


def test_struct_layouts_match_the_header():
    # sizes the C compiler gives the boundary structs (x86-64 SysV): a drifted ctypes mirror would corrupt calls
    N = rtc._native
    assert ctypes.sizeof(N.Material) == 3 * 8 + 7 * 8 + 2 * 4
    assert ctypes.sizeof(N.RenderOpts) == 16 * 4
    assert ctypes.sizeof(N.Stats) == 8 * (7 + 7 + 7 + 10) + 2 * 8
    assert ctypes.sizeof(N.SceneInfo) == 14 * 4 + 8 + 2 * 4
    o = N.RenderOpts()
    N.lib().rtc_render_opts_default(ctypes.byref(o))
    assert (o.depth, o.tile_count, o.use_bvh, o.collect_stats, o.schedule) == (5, 1, 0, 0, 0)  # camera.rs:80
    m = N.Material()
    N.lib().rtc_material_default(ctypes.byref(m))  # material.rs:83-96
    assert list(m.color) == [1.0, 1.0, 1.0] and (m.ambient, m.diffuse, m.specular, m.shininess) == (0.1, 0.9, 0.9, 200.0)
    assert (m.reflectivity, m.transparency, m.refractive_index, m.casts_shadow, m.pattern) == (0.0, 0.0, 1.0, 1, -1)


def test_struct_layouts_match_what_a_c_compiler_sees(tmp_path):
    # include/rtc.h through gcc (plain C, as a cgo / bindgen / ctypes user would consume it): sizes and a few offsets
    import subprocess
    N = rtc._native
    src = tmp_path / "sizes.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "rtc.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(rtc_material), sizeof(rtc_render_opts), sizeof(rtc_stats),
           sizeof(rtc_scene_info), sizeof(rtc_item_info), sizeof(rtc_light_info), sizeof(rtc_obj_info),
           offsetof(rtc_stats, kernel_ms));
    printf("%zu %zu %zu %zu\\n", offsetof(rtc_render_opts, exchange), offsetof(rtc_render_opts, fp_mode),
           offsetof(rtc_stats, shadow_rays_skipped), offsetof(rtc_render_opts, tail_polls));
    return 0;
}
''')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    a, b = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    sizes = [int(x) for x in a.split()]
    assert sizes == [ctypes.sizeof(N.Material), ctypes.sizeof(N.RenderOpts), ctypes.sizeof(N.Stats),
                     ctypes.sizeof(N.SceneInfo), ctypes.sizeof(N.ItemInfo), ctypes.sizeof(N.LightInfo),
                     ctypes.sizeof(N.ObjInfo), N.Stats.kernel_ms.offset]
    assert [int(x) for x in b.split()] == [N.RenderOpts.exchange.offset, N.RenderOpts.fp_mode.offset,
                                           N.Stats.shadow_rays_skipped.offset, N.RenderOpts.tail_polls.offset]
