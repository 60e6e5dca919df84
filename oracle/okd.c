/*
 * oracle/okd.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 * 2-D restatement of the reference kd-tree semantics; see okd.h for the
 * file:line map into cpp/trg_planner/core/trg_planner/src/kdtree/kdtree.c.
 */
#include "okd.h"

#include <math.h>
#include <stdlib.h>

typedef struct onode {
  float *xy;            /* separately allocated, as the reference does (kdtree.c:176) */
  int axis;             /* 0 = split on x, 1 = split on y */
  void *payload;
  struct onode *lo, *hi; /* lo: coordinate < this, hi: coordinate >= this */
} onode;

typedef struct oitem {
  onode *n;
  struct oitem *next;
} oitem;

struct okdtree {
  onode *root;
  int have_box;
  float bmin[2], bmax[2]; /* bounding box of everything inserted (kdtree.c:202-206) */
  int depth;
};

struct okdres {
  oitem *head; /* sentinel */
  oitem *it;
  int size;
};

struct okdtree *okd_create(void) {
  struct okdtree *t = (struct okdtree *)malloc(sizeof *t);
  if (!t) return 0;
  t->root = 0;
  t->have_box = 0;
  t->depth = 0;
  return t;
}

static void drop_subtree(onode *n) {
  if (!n) return;
  drop_subtree(n->lo);
  drop_subtree(n->hi);
  free(n->xy);
  free(n);
}

void okd_clear(struct okdtree *t) {
  drop_subtree(t->root);
  t->root = 0;
  t->have_box = 0;
  t->depth = 0;
}

void okd_free(struct okdtree *t) {
  if (!t) return;
  okd_clear(t);
  free(t);
}

int okd_depth(struct okdtree *t) { return t->depth; }

/* kdtree.c:167-194 -- descend until an empty slot; `<` goes to the low side. */
int okd_insert2(struct okdtree *t, float x, float y, void *data) {
  float p[2];
  onode **slot = &t->root;
  int axis = 0, level = 1;
  p[0] = x;
  p[1] = y;
  while (*slot) {
    onode *cur = *slot;
    slot = (p[cur->axis] < cur->xy[cur->axis]) ? &cur->lo : &cur->hi;
    axis = (cur->axis + 1) % 2;
    ++level;
  }
  onode *n = (onode *)malloc(sizeof *n);
  if (!n) return -1;
  n->xy = (float *)malloc(2 * sizeof(float));
  if (!n->xy) {
    free(n);
    return -1;
  }
  n->xy[0] = x;
  n->xy[1] = y;
  n->axis = axis;
  n->payload = data;
  n->lo = n->hi = 0;
  *slot = n;
  if (level > t->depth) t->depth = level;

  /* kdtree.c:202-206, 679-691 */
  if (!t->have_box) {
    t->bmin[0] = t->bmax[0] = x;
    t->bmin[1] = t->bmax[1] = y;
    t->have_box = 1;
  } else {
    for (int i = 0; i < 2; ++i) {
      if (p[i] < t->bmin[i]) t->bmin[i] = p[i];
      if (p[i] > t->bmax[i]) t->bmax[i] = p[i];
    }
  }
  return 0;
}

static struct okdres *res_new(void) {
  struct okdres *r = (struct okdres *)malloc(sizeof *r);
  if (!r) return 0;
  r->head = (oitem *)malloc(sizeof(oitem)); /* sentinel, kdtree.c:379,487 */
  if (!r->head) {
    free(r);
    return 0;
  }
  r->head->next = 0;
  r->it = 0;
  r->size = 0;
  return r;
}

/* unordered insert = push right behind the sentinel (kdtree.c:759-777, dist_sq=-1) */
static int res_push_front(struct okdres *r, onode *n) {
  oitem *it = (oitem *)malloc(sizeof *it);
  if (!it) return -1;
  it->n = n;
  it->next = r->head->next;
  r->head->next = it;
  return 0;
}

/* kdtree.c:270-301 */
static int range_walk(onode *n, const float *q, float range, struct okdres *r) {
  if (!n) return 0;
  int added = 0;
  float d2 = 0;
  for (int i = 0; i < 2; ++i) d2 += (n->xy[i] - q[i]) * (n->xy[i] - q[i]);
  if (d2 <= range * range) {
    if (res_push_front(r, n) < 0) return -1;
    added = 1;
  }
  float dx = q[n->axis] - n->xy[n->axis];
  int ret = range_walk(dx <= 0.0 ? n->lo : n->hi, q, range, r);
  if (ret >= 0 && fabs(dx) < range) {
    added += ret;
    ret = range_walk(dx <= 0.0 ? n->hi : n->lo, q, range, r);
  }
  if (ret < 0) return -1;
  return added + ret;
}

struct okdres *okd_nearest_range2(struct okdtree *t, float x, float y, float range) {
  float q[2] = {x, y};
  struct okdres *r = res_new();
  if (!r) return 0;
  int n = range_walk(t->root, q, range, r);
  if (n < 0) {
    okd_res_free(r);
    return 0;
  }
  r->size = n;
  r->it = r->head->next;
  return r;
}

/* kdtree.c:693-707 */
static float box_d2(const float *bmin, const float *bmax, const float *q) {
  float acc = 0;
  for (int i = 0; i < 2; ++i) {
    if (q[i] < bmin[i]) {
      acc += (bmin[i] - q[i]) * (bmin[i] - q[i]);
    } else if (q[i] > bmax[i]) {
      acc += (bmax[i] - q[i]) * (bmax[i] - q[i]);
    }
  }
  return acc;
}

/* kdtree.c:303-362: nearer child (box sliced), then self, then farther child if its box can win */
static void nn_walk(onode *n, const float *q, onode **best, float *best_d2, float *bmin, float *bmax) {
  int ax = n->axis;
  onode *nearer, *farther;
  float *near_edge, *far_edge;
  if (q[ax] - n->xy[ax] <= 0) {
    nearer = n->lo;
    farther = n->hi;
    near_edge = bmax + ax;
    far_edge = bmin + ax;
  } else {
    nearer = n->hi;
    farther = n->lo;
    near_edge = bmin + ax;
    far_edge = bmax + ax;
  }
  if (nearer) {
    float keep = *near_edge;
    *near_edge = n->xy[ax];
    nn_walk(nearer, q, best, best_d2, bmin, bmax);
    *near_edge = keep;
  }
  float d2 = 0;
  for (int i = 0; i < 2; ++i) d2 += (n->xy[i] - q[i]) * (n->xy[i] - q[i]);
  if (d2 < *best_d2) {
    *best = n;
    *best_d2 = d2;
  }
  if (farther) {
    float keep = *far_edge;
    *far_edge = n->xy[ax];
    if (box_d2(bmin, bmax, q) < *best_d2) nn_walk(farther, q, best, best_d2, bmin, bmax);
    *far_edge = keep;
  }
}

/* kdtree.c:364-417 */
struct okdres *okd_nearest2(struct okdtree *t, float x, float y) {
  if (!t || !t->have_box) return 0;
  float q[2] = {x, y};
  struct okdres *r = res_new();
  if (!r) return 0;
  /* the reference duplicates the bounding box on the heap per call (kdtree.c:387) */
  float *box = (float *)malloc(4 * sizeof(float));
  if (!box) {
    okd_res_free(r);
    return 0;
  }
  box[0] = t->bmin[0];
  box[1] = t->bmin[1];
  box[2] = t->bmax[0];
  box[3] = t->bmax[1];
  onode *best = t->root;
  float best_d2 = 0;
  for (int i = 0; i < 2; ++i) best_d2 += (best->xy[i] - q[i]) * (best->xy[i] - q[i]);
  nn_walk(t->root, q, &best, &best_d2, box, box + 2);
  free(box);
  if (res_push_front(r, best) < 0) {
    okd_res_free(r);
    return 0;
  }
  r->size = 1;
  r->it = r->head->next;
  return r;
}

void okd_res_free(struct okdres *r) {
  oitem *it = r->head->next;
  while (it) {
    oitem *nx = it->next;
    free(it);
    it = nx;
  }
  free(r->head);
  free(r);
}

int okd_res_size(struct okdres *r) { return r->size; }
int okd_res_end(struct okdres *r) { return r->it == 0; }
int okd_res_next(struct okdres *r) {
  r->it = r->it->next;
  return r->it != 0;
}
void *okd_res_item_data(struct okdres *r) { return r->it ? r->it->n->payload : 0; }
