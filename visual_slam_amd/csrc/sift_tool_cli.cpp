// vsm_sift_tool: the two loops of the reference's converter program (test_main.cpp:30-60) over the C ABI.
//   vsm_sift_tool txt2sift IN_DIR OUT_DIR N     <IN_DIR>/<i>.txt  -> <OUT_DIR>/<i>.sift  for i in 0..N-1
//   vsm_sift_tool sift2txt IN_DIR OUT_DIR N [locations|descriptors]   <IN_DIR>/<i>.sift -> <OUT_DIR>/<i>.txt
// and FeatureData::ReadSIFTA + ConvertA2B (FeaturePoints.cpp:54-72, 655-710) for ASCII feature files:
//   vsm_sift_tool a2b FILE...        x.sift.sift -> x.sift (ASCII file removed), x.sift converted in place
#include "vsm.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, char** argv)
{
    if (argc >= 3 && strcmp(argv[1], "a2b") == 0) {
        long long total = 0;
        for (int i = 2; i < argc; ++i) {
            char name[4096];
            const int rc = vsm_sift_ascii_to_binary(argv[i], name, (int)sizeof name);
            if (rc < 0) {
                fprintf(stderr, "%s: error %d\n", argv[i], rc);
                return 1;
            }
            if (!name[0]) fprintf(stderr, "%s: not a .sift name, nothing written\n", argv[i]);
            total += rc;
        }
        fprintf(stderr, "%d files, %lld features\n", argc - 2, total);
        return 0;
    }
    if (argc < 5 || (strcmp(argv[1], "txt2sift") != 0 && strcmp(argv[1], "sift2txt") != 0)) {
        fprintf(stderr, "usage: vsm_sift_tool txt2sift|sift2txt IN_DIR OUT_DIR N [locations|descriptors]\n"
                        "       vsm_sift_tool a2b FILE...\n");
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
