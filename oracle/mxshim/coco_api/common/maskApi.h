/* maskApi.h -- declarations of the four COCO mask-API functions ProposalMaskTarget calls.
 *
 * TEST INFRASTRUCTURE ONLY.  The reference includes "../coco_api/common/maskApi.h" from
 * github.com/RogerChern/cocoapi (doc/INSTALL.md:90-93), which is NOT vendored in /root/reference.
 * The definitions are in oracle/mask_api.c: a restatement of the published pycocotools algorithm
 * (common/maskApi.c, Piotr Dollar & Tsung-Yi Lin, 2014).  => the mask rasterisation is
 * "parity unpinned"; everything around it in proposal_mask_target.cc is the reference's own code.
 */
#ifndef ORACLE_MASK_API_H_
#define ORACLE_MASK_API_H_
#ifdef __cplusplus
extern "C" {
#endif
typedef unsigned int uint;
typedef unsigned long siz;
typedef unsigned char byte;
typedef struct { siz h, w, m; uint *cnts; } RLE;

void rlesInit(RLE **R, siz n);
void rlesFree(RLE **R, siz n);
void rleFrPoly(RLE *R, const double *xy, siz k, siz h, siz w);
void rleDecode(const RLE *R, byte *mask, siz n);
#ifdef __cplusplus
}
#endif
#endif
