/*
 * oracle/okd.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the semantics of the reference's spatial index, the
 * vendored "kdtree" library (unbalanced k-d tree, float coordinates):
 *   reference: cpp/trg_planner/core/trg_planner/src/kdtree/kdtree.c
 *              cpp/trg_planner/core/trg_planner/include/kdtree/kdtree.h:37-111
 *
 * Only the 2-D entry points the TRG path uses are restated:
 *   okd_insert2        <- kd_insert2 / kd_insert / insert_rec   (kdtree.c:244-250,196-209,167-194)
 *   okd_nearest2       <- kd_nearest2 / kd_nearest / kd_nearest_i (kdtree.c:453-459,364-417,303-362)
 *   okd_nearest_range2 <- kd_nearest_range2 / find_nearest       (kdtree.c:537-543,479-501,270-301)
 *   okd_res_*          <- kd_res_* iterator                      (kdtree.c:563-639)
 *
 * Behaviour kept on purpose (it changes results or cost):
 *   - go left iff pos[dir] < node->pos[dir], alternate x/y          (kdtree.c:189-193)
 *   - range test is inclusive `<=` in fp32, accumulated x then y    (kdtree.c:277-281)
 *   - far side visited iff fabs(dx) < range                         (kdtree.c:291)
 *   - unordered results are pushed at the list head, so iteration is
 *     REVERSE discovery order                                        (kdtree.c:282, 759-777)
 *   - 1-NN starts from the root as best, updates on strict `<`,
 *     order nearer subtree -> self -> farther subtree, farther pruned
 *     by hyperrect_dist_sq(rect) < best                              (kdtree.c:327-361, 693-707)
 *   - one malloc per node + one for its coordinates, one malloc per
 *     result item (that allocation pattern IS the reference's cost)  (kdtree.c:173-176, 763)
 *
 * Pinning: cross-checked against the reference kdtree.c itself, compiled in
 * place into oracle/_ref/ (see oracle/Makefile, tests/test_oracle_kd.py).
 */
#ifndef ORACLE_OKD_H_
#define ORACLE_OKD_H_

#ifdef __cplusplus
extern "C" {
#endif

struct okdtree;
struct okdres;

struct okdtree *okd_create(void);
void okd_free(struct okdtree *t);
void okd_clear(struct okdtree *t);
int okd_insert2(struct okdtree *t, float x, float y, void *data);
struct okdres *okd_nearest2(struct okdtree *t, float x, float y);
struct okdres *okd_nearest_range2(struct okdtree *t, float x, float y, float range);
void okd_res_free(struct okdres *r);
int okd_res_size(struct okdres *r);
int okd_res_end(struct okdres *r);
int okd_res_next(struct okdres *r);
void *okd_res_item_data(struct okdres *r);
/* instrumentation: deepest node level reached by inserts so far */
int okd_depth(struct okdtree *t);

#ifdef __cplusplus
}
#endif
#endif
