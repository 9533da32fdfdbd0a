#include <hdf5.h>
#include <string.h>
#include <stdlib.h>
/* make_keras_h5.c -- TEST FIXTURE GENERATOR (not product code): writes a Keras-layout weight file with the REAL HDF5 library
   the way h5py does it (default property lists = libver "earliest"; argv[8] = "latest" switches to the newest format):
   /<layer>/<model>/<layer>/kernel:0, bias:0; attrs layer_names, backend, keras_version at root; weight_names per layer
   (the layout of Keras' save_weights_to_hdf5_group for a subclassed model of Dense layers, reference nerf.py:63-64).
   Values: v[i] = ((lcg >> 40) & 0xFFFF) / 65536 - 0.5 with lcg = lcg * 6364136223846793005 + 1442695040888963407, seeded 12345,
   drawn in file order (per layer: kernel then bias); tests/test_hdf5_min.py regenerates them.
   usage: make_keras_h5 out.h5 model_name n_layers units skip xyz_dim dir_dim [latest]
   build: gcc -o make_keras_h5 make_keras_h5.c -I<hdf5>/include -L<hdf5>/lib -lhdf5 */
static void str_attr_array(hid_t loc, const char* name, const char** vals, int n) {
    size_t maxlen = 1; for (int i = 0; i < n; ++i) if (strlen(vals[i]) > maxlen) maxlen = strlen(vals[i]);
    hid_t t = H5Tcopy(H5T_C_S1); H5Tset_size(t, maxlen); H5Tset_strpad(t, H5T_STR_NULLPAD);
    hsize_t d = n; hid_t s = H5Screate_simple(1, &d, NULL);
    char* buf = calloc(n, maxlen); for (int i = 0; i < n; ++i) memcpy(buf + i * maxlen, vals[i], strlen(vals[i]));
    hid_t a = H5Acreate2(loc, name, t, s, H5P_DEFAULT, H5P_DEFAULT); H5Awrite(a, t, buf); H5Aclose(a); H5Sclose(s); H5Tclose(t); free(buf);
}
static void vlen_attr(hid_t loc, const char* name, const char* val) {
    hid_t t = H5Tcopy(H5T_C_S1); H5Tset_size(t, H5T_VARIABLE);
    hid_t s = H5Screate(H5S_SCALAR); hid_t a = H5Acreate2(loc, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, &val); H5Aclose(a); H5Sclose(s); H5Tclose(t);
}
int main(int argc, char** argv) {
    const char* model = argv[2];
    int n_layers = atoi(argv[3]), units = atoi(argv[4]), skip = atoi(argv[5]), xyz = atoi(argv[6]), dir = atoi(argv[7]);
    hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
    if (argc > 8 && !strcmp(argv[8], "latest")) H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST);
    hid_t f = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
    int nl = n_layers + 4; const char** names = malloc(nl * sizeof(char*)); int* fi = malloc(nl * 4); int* fo = malloc(nl * 4);
    int in = xyz;
    for (int i = 0; i < n_layers; ++i) { char* b = malloc(32); sprintf(b, "layer_%d", i); names[i] = b; fi[i] = in; fo[i] = units; in = units; if (i % skip == 0 && i > 0) in = units + xyz; }
    names[n_layers] = "sigma"; fi[n_layers] = in; fo[n_layers] = 1;
    names[n_layers + 1] = "features"; fi[n_layers + 1] = in; fo[n_layers + 1] = units;
    names[n_layers + 2] = "rgb_features"; fi[n_layers + 2] = units + dir; fo[n_layers + 2] = units / 2;
    names[n_layers + 3] = "rgb"; fi[n_layers + 3] = units / 2; fo[n_layers + 3] = 3;
    str_attr_array(f, "layer_names", names, nl);
    vlen_attr(f, "backend", "tensorflow"); vlen_attr(f, "keras_version", "2.9.0");
    unsigned long long seed = 12345;
    for (int l = 0; l < nl; ++l) {
        hid_t g = H5Gcreate2(f, names[l], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        char wk[128], wb[128]; sprintf(wk, "%s/%s/kernel:0", model, names[l]); sprintf(wb, "%s/%s/bias:0", model, names[l]);
        const char* wn[2] = {wk, wb}; str_attr_array(g, "weight_names", wn, 2);
        hid_t lcpl = H5Pcreate(H5P_LINK_CREATE); H5Pset_create_intermediate_group(lcpl, 1);
        for (int w = 0; w < 2; ++w) {
            hsize_t dims[2] = {(hsize_t)fi[l], (hsize_t)fo[l]}; int rank = w == 0 ? 2 : 1; if (w == 1) dims[0] = fo[l];
            size_t n = w == 0 ? (size_t)fi[l] * fo[l] : fo[l];
            float* v = malloc(n * 4);
            for (size_t i = 0; i < n; ++i) { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; v[i] = (float)((seed >> 40) & 0xFFFF) / 65536.0f - 0.5f; }
            hid_t s = H5Screate_simple(rank, dims, NULL);
            hid_t d = H5Dcreate2(g, wn[w], H5T_IEEE_F32LE, s, lcpl, H5P_DEFAULT, H5P_DEFAULT);
            H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, v); H5Dclose(d); H5Sclose(s); free(v);
        }
        H5Pclose(lcpl); H5Gclose(g);
    }
    hid_t g = H5Gcreate2(f, "top_level_model_weights", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT); H5Gclose(g);
    H5Fclose(f); return 0;
}
