// vsm_sift_tool: the two loops of the reference's converter program (test_main.cpp:30-60) over the C ABI.
//   vsm_sift_tool txt2sift IN_DIR OUT_DIR N     <IN_DIR>/<i>.txt  -> <OUT_DIR>/<i>.sift  for i in 0..N-1
//   vsm_sift_tool sift2txt IN_DIR OUT_DIR N [locations|descriptors]   <IN_DIR>/<i>.sift -> <OUT_DIR>/<i>.txt
#include "vsm.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, char** argv)
{
    if (argc < 5 || (strcmp(argv[1], "txt2sift") != 0 && strcmp(argv[1], "sift2txt") != 0)) {
        fprintf(stderr, "usage: vsm_sift_tool txt2sift|sift2txt IN_DIR OUT_DIR N [locations|descriptors]\n");
        return 2;
    }
    const bool to_sift = strcmp(argv[1], "txt2sift") == 0;
    const std::string in = argv[2], out = argv[3];
    const int n = atoi(argv[4]);
    int what = VSM_TXT_BOTH;
    if (argc > 5) what = strcmp(argv[5], "locations") == 0 ? VSM_TXT_LOCATIONS : strcmp(argv[5], "descriptors") == 0 ? VSM_TXT_DESCRIPTORS : -1;
    if (what < 0 || n < 0) {
        fprintf(stderr, "bad argument\n");
        return 2;
    }
    long long features = 0;
    for (int i = 0; i < n; ++i) {
        const std::string a = in + "/" + std::to_string(i) + (to_sift ? ".txt" : ".sift");
        const std::string b = out + "/" + std::to_string(i) + (to_sift ? ".sift" : ".txt");
        const int rc = to_sift ? vsm_txt_to_sift(a.c_str(), b.c_str()) : vsm_sift_to_txt(a.c_str(), b.c_str(), what);
        if (rc < 0) {
            fprintf(stderr, "%s -> %s: error %d\n", a.c_str(), b.c_str(), rc);
            return 1;
        }
        features += rc;
    }
    fprintf(stderr, "%d files, %lld features\n", n, features);
    return 0;
}
