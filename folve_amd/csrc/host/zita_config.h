// This file restates the interface and behaviour of folve's zita-config.h (itself derived from jconvolver's config.h, Copyright (C) 2006-2011 Fons Adriaensen, GPL v2 or later) and, Copyright (C) 2012 Henner Zeller
// <h.zeller@acm.org>, free software under the GNU General Public License, version 3 or (at your option) any later
// version; this restatement is distributed under the same terms, WITHOUT ANY WARRANTY (<http://www.gnu.org/licenses/>).
// zita_config.h — jconvolver-format filter configuration loader.
//
// Behavioural mirror of the reference loader with `Convproc*` replaced by the
// engine's filter handle:
//   struct ZitaConfig, error enum, MAXSIZE   zita-config.h:37-61
//   config()                                  zita-config.cc:282-378
//   convnew / inpname / outname               zita-fconfig.cc:38-109
//   readfile / impdirac / imphilbert / impcopy zita-config.cc:55-279
// Same grammar, same error codes, same "ERR_OTHER is swallowed" rule
// (zita-config.cc:345).  Diagnostics go to the log hook instead of syslog.
#pragma once

#include "../../../include/folve_engine.h"

namespace folve {

struct ZitaConfig {
    const char* config_file;   // configuration file we're reading from
    fe_engine* engine;         // GPU the filter will live on (may be NULL: assemble only)
    fe_filter* filter;         // resulting filter object (was: Convproc *convproc)

    // Parameters (zita-config.h:41-48).
    int latency;
    int options;
    int fsamp;
    int fragm;
    int ninp;
    int nout;
    int size;

    // Not in the reference: called with the resolved path of every impulse file /impulse/read opens, so that the
    // caller can implement the reference's own TODO — "this should as well check if any *.wav file mentioned is still
    // the same timestamp" (sound-processor.cc:129-133).  NULL: nobody listens.
    void (*on_impulse_file)(void* user, const char* path);
    void* on_impulse_user;
};

enum { NOERR, ERR_OTHER, ERR_SYNTAX, ERR_PARAM, ERR_ALLOC, ERR_CANTCD, ERR_COMMAND, ERR_NOCONV, ERR_IONUM };

#define FOLVE_MAXSIZE 0x00100000

// Returns 0 on success, -1 if the file cannot be opened, else one of the ERR_*
// codes above (never ERR_OTHER).  Does not commit the filter.
int config(ZitaConfig* cfg, const char* config_file);
int convnew(ZitaConfig* cfg, const char* line, int lnum);
int inpname(ZitaConfig* cfg, const char* line);
int outname(ZitaConfig* cfg, const char* line);

// Diagnostics hook (the reference uses syslog(LOG_ERR, ...)). Default: silent
// unless FOLVE_AMD_LOG=1, then stderr.
typedef void (*LogFn)(const char* msg);
void SetLogHandler(LogFn fn);
void Logf(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

}  // namespace folve
