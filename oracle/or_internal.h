/* or_internal.h -- ORACLE (test infrastructure): helpers shared between oracle translation units */
#ifndef LUT_LDPC_ORACLE_INTERNAL_H
#define LUT_LDPC_ORACLE_INTERNAL_H
#include "oracle.h"

or_ivec or__design_skip_zero_mass(or_dvec *p_out, or_dvec prod, int Nq);

/* minimal INI reader (sections, key = value, ';' and '#' comment lines), standing in for
 * boost::property_tree::ini_parser as used at src/LDPC_BER_Sim.cpp:50 and src/LDPC_DE.cpp:1149 */
typedef struct { char *section, *key, *value; } or_ini_entry;
typedef struct { or_ini_entry *e; int n; } or_ini;
or_ini     *or_ini_load(const char *path);
void        or_ini_free(or_ini *ini);
const char *or_ini_get(const or_ini *ini, const char *section, const char *key); /* NULL if absent */
int         or_ini_has_section(const or_ini *ini, const char *section);

#endif
