/* tables.h - the active name tables, for the library's own translation units */
#ifndef MCX_TABLES_H
#define MCX_TABLES_H

#include <stdint.h>

#include "../../include/mcx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MCX_TABLES_FORMAT 1
enum { MCX_TS_CORE, MCX_TS_PREFIX, MCX_TS_HW_PREFIX, MCX_TS_INFRA, MCX_TS_VARIANT, MCX_TS_COUNT };

typedef struct {
    int format;
    const char *const *item[MCX_TS_COUNT]; /* core names sorted; variants as (prefix, replacement) pairs */
    uint32_t count[MCX_TS_COUNT];          /* entries of item[s] (variants: two per rule) */
} mcx_tables_view;

const mcx_tables_view *mcx_tables_active(void);
/* grows by one whenever the active set is replaced */
uint64_t mcx_tables_generation(void);

#ifdef __cplusplus
}
#endif
#endif /* MCX_TABLES_H */
