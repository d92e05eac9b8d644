/* TEST INFRASTRUCTURE (CPU only): include/gnark_backend.h compiles as C, its structures have cgo's layout, and libgnark_backend.so resolves every one of
 * the ten names with these prototypes (the two that upstream answers `false` without looking at anything are called; no other call: they need a GPU). */
#include <stdio.h>
#include <string.h>
#include "gnark_backend.h"
int main(void) {
    GoString g = {"{}", 2};
    if (sizeof(GoString) != 2 * sizeof(void *) || sizeof(KeyPair) != 2 * sizeof(void *) || sizeof(GoUint8) != 1) return 2;
    void *fns[] = {(void *)PlonkProveWithPK, (void *)PlonkVerifyWithMeta, (void *)PlonkVerifyWithVK, (void *)PlonkPreprocess, (void *)PlonkProveWithMeta,
                   (void *)ProveWithMeta, (void *)ProveWithPK, (void *)VerifyWithMeta, (void *)VerifyWithVK, (void *)Preprocess};
    for (unsigned i = 0; i < sizeof fns / sizeof *fns; i++)
        if (!fns[i]) return 3;
    if (PlonkVerifyWithMeta(g, g, g) != 0 || VerifyWithMeta(g, g) != 0) return 4;
    printf("ten exports resolved\n");
    return 0;
}
