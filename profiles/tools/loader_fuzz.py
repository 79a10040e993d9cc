"""3000 mangled scene files through vcm_scene_load (the cases of tests/test_scene2.py::test_the_loader_survives_mangled_files, more of
them), meant to run against the sanitizer build: profiles/tools/loader_asan.sh.   python loader_fuzz.py [seed]"""
import ctypes as C, os, random, sys
L = C.CDLL(os.environ.get('SCENE_LIB', '/tmp/asan/libscene_asan.so'))
L.vcm_scene_load.restype = C.c_void_p
L.vcm_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
L.vcm_scene_file_free.argtypes = [C.c_void_p]
L.vcm_scene_load_error.restype = C.c_char_p
work = os.environ.get('WORK', '/tmp/asan/w'); os.makedirs(work, exist_ok=True)
base = {
 "s.vcmscene": "obj m.obj\nsphere 0.2 0.2 0.5 0.25 mirror\ncamera 0 -4 0.2 0 1 0 0 0 1 45\n",
 "m.obj": "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nusemtl white\nf 1 2 3 4\nv 0 0 2\nv 1 0 2\nv 0 1 2\nusemtl lamp\nf -3 -2 -1\nusemtl glossy\nf 1/1/1 2/2/2 3/3/3\n",
 "m.mtl": "newmtl white\nKd 0.8 0.8 0.8\nnewmtl glossy\nKd 0.1 0.1 0.1\nKs 0.7 0.7 0.7\nNs 90\nnewmtl mirror\nKs 1 1 1\nillum 3\nnewmtl lamp\nKe 25 25 25\n",
}
junk = ["", "-1", "0", "1e39", "-1e39", "nan", "inf", "99999999999999999999", "x", "/", "//", "1/", "#", "\x00", "-0", "3000000000", "f", "v", "usemtl", "\t", "-2147483648", "2147483647", "1e-320"]
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
loaded = refused = 0
for case in range(3000):
    files = dict(base)
    name = rng.choice(list(files))
    text = files[name]
    how = rng.randrange(6)
    if how == 0:
        b = bytearray(text.encode())
        for _ in range(rng.randrange(1, 6)): b[rng.randrange(len(b))] = rng.randrange(256)
        data = bytes(b)
    else:
        lines = text.split("\n")
        i = rng.randrange(len(lines))
        toks = lines[i].split(" ")
        if how == 1 and toks: toks[rng.randrange(len(toks))] = rng.choice(junk)
        elif how == 2: toks = toks[:rng.randrange(len(toks) + 1)]
        elif how == 3: lines.insert(i, lines[i]); toks = lines[i].split(" ")
        elif how == 4: toks = []
        elif how == 5: toks = toks + [rng.choice(junk)] * rng.randrange(1, 40)
        lines[i] = " ".join(toks)
        data = "\n".join(lines).encode()
        if rng.random() < 0.2: data = data[:rng.randrange(len(data) + 1)]
    for n, t in files.items():
        open(os.path.join(work, n), "wb").write(data if n == name else t.encode())
    h = L.vcm_scene_load(os.path.join(work, "s.vcmscene").encode(), 16, 16)
    if h: loaded += 1; L.vcm_scene_file_free(h)
    else: refused += 1
print("loaded", loaded, "refused", refused)
