// declaration-only stand-in for popt 1.16
#pragma once
#define POPT_ARG_NONE 0U
#define POPT_ARG_STRING 1U
#define POPT_ARG_INT 2U
#define POPT_ARG_INCLUDE_TABLE 4U
struct poptOption { const char* longName; char shortName; unsigned int argInfo; void* arg; int val; const char* descrip; const char* argDescrip; };
extern struct poptOption poptHelpOptions[];
#define POPT_AUTOHELP {nullptr, '\0', POPT_ARG_INCLUDE_TABLE, poptHelpOptions, 0, "Help options:", nullptr},
typedef struct poptContext_s* poptContext;
extern "C" {
poptContext poptGetContext(const char* name, int argc, const char** argv, const struct poptOption* options, unsigned int flags);
int poptGetNextOpt(poptContext con);
poptContext poptFreeContext(poptContext con);
}
