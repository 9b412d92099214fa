/* or_ini.c -- ORACLE (test infrastructure): minimal INI reader, see or_internal.h */
#define _GNU_SOURCE
#include "or_internal.h"
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char *trim_dup(const char *b, const char *e)
{
    while (b < e && isspace((unsigned char)*b)) b++;
    while (e > b && isspace((unsigned char)e[-1])) e--;
    return strndup(b, (size_t)(e - b));
}

or_ini *or_ini_load(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    or_ini *ini = (or_ini *)calloc(1, sizeof(or_ini));
    char *line = NULL, *section = strdup(""); size_t cap = 0; ssize_t len;
    int has_sections_cap = 0;
    while ((len = getline(&line, &cap, f)) >= 0) {
        const char *b = line, *e = line + len;
        while (b < e && isspace((unsigned char)*b)) b++;
        while (e > b && isspace((unsigned char)e[-1])) e--;
        if (b == e || *b == ';' || *b == '#') continue;
        if (*b == '[') {
            const char *r = memchr(b, ']', (size_t)(e - b));
            if (!r) continue;
            free(section); section = trim_dup(b + 1, r);
            /* remember bare sections so that has_section works for key-less ones */
            if (ini->n >= has_sections_cap) { has_sections_cap = has_sections_cap ? has_sections_cap * 2 : 32; ini->e = (or_ini_entry *)realloc(ini->e, sizeof(or_ini_entry) * (size_t)has_sections_cap); }
            ini->e[ini->n].section = strdup(section); ini->e[ini->n].key = strdup(""); ini->e[ini->n].value = strdup(""); ini->n++;
            continue;
        }
        const char *eq = memchr(b, '=', (size_t)(e - b));
        if (!eq) continue;
        if (ini->n >= has_sections_cap) { has_sections_cap = has_sections_cap ? has_sections_cap * 2 : 32; ini->e = (or_ini_entry *)realloc(ini->e, sizeof(or_ini_entry) * (size_t)has_sections_cap); }
        ini->e[ini->n].section = strdup(section);
        ini->e[ini->n].key = trim_dup(b, eq);
        ini->e[ini->n].value = trim_dup(eq + 1, e);
        ini->n++;
    }
    free(line); free(section); fclose(f);
    return ini;
}

void or_ini_free(or_ini *ini)
{
    if (!ini) return;
    for (int i = 0; i < ini->n; i++) { free(ini->e[i].section); free(ini->e[i].key); free(ini->e[i].value); }
    free(ini->e); free(ini);
}

const char *or_ini_get(const or_ini *ini, const char *section, const char *key)
{
    for (int i = 0; i < ini->n; i++)
        if (ini->e[i].key[0] && !strcmp(ini->e[i].section, section) && !strcmp(ini->e[i].key, key)) return ini->e[i].value;
    return NULL;
}

int or_ini_has_section(const or_ini *ini, const char *section)
{
    for (int i = 0; i < ini->n; i++) if (!strcmp(ini->e[i].section, section)) return 1;
    return 0;
}
