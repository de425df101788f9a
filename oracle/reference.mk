# oracle/reference.mk — how the reference is compiled in place (included by oracle/Makefile and integration/Makefile): where its sources lie, the flags of
# its own Release build, the object lists.  Reference objects are built ONLY by oracle/Makefile, into oracle/_ref/obj{8,10,12}/.
REFMK_DIR := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
REF      ?= /root/reference
SRC      := $(REF)/source
OUT      := $(REFMK_DIR)_ref
CXX      ?= g++
CC       ?= gcc

LIBSRCS  := $(filter-out $(SRC)/common/winxp.cpp,$(wildcard $(SRC)/common/*.cpp)) $(wildcard $(SRC)/encoder/*.cpp)
CLISRCS  := $(wildcard $(SRC)/input/*.cpp) $(wildcard $(SRC)/output/*.cpp) $(SRC)/x265.cpp $(SRC)/x265cli.cpp $(SRC)/abrEncApp.cpp

COMMON_DEFS := -DEXPORT_C_API=1 -DX265_ARCH_X86=1 -DX86_64=1 -DHAVE_INT_TYPES_H=1 -DHAVE_STRTOK_R=1 \
               -DENABLE_LIBNUMA=0 -DX265_VERSION=3.4+28-ref -D__STDC_LIMIT_MACROS=1
INCS     := -I$(OUT) -I$(SRC) -I$(SRC)/common -I$(SRC)/encoder -I$(SRC)/input -I$(SRC)/output
# The reference's own default build is CMake Release on GCC (source/CMakeLists.txt:2-7): -O3 -DNDEBUG, plus what its GCC branch adds
# (:209-319): -ffast-math -mstackrealign -fno-exceptions.  REF_OPT is used for EVERY reference object (the CPU baseline, the pinned
# oracle library and the objects the bound encoder links); HOST_OPT for the binding's own TUs (no -ffast-math / -fno-exceptions there:
# they hold float expressions that must stay IEEE and they use the C++ runtime).
REF_OPT  ?= -O3 -DNDEBUG -ffast-math -mstackrealign -fno-exceptions
HOST_OPT ?= -O3 -DNDEBUG
CXXFLAGS_REF  := $(REF_OPT) -std=gnu++11 -fPIC -w $(COMMON_DEFS) $(INCS)
CXXFLAGS_HOST := $(HOST_OPT) -std=gnu++11 -fPIC -w $(COMMON_DEFS) $(INCS)

DEFS8    := -DX265_DEPTH=8  -DHIGH_BIT_DEPTH=0 -DX265_NS=x265
DEFS10   := -DX265_DEPTH=10 -DHIGH_BIT_DEPTH=1 -DX265_NS=x265
DEFS12   := -DX265_DEPTH=12 -DHIGH_BIT_DEPTH=1 -DX265_NS=x265

obj = $(patsubst $(SRC)/%.cpp,$(OUT)/obj$(1)/%.o,$(2))

LIBOBJS8  := $(call obj,8,$(LIBSRCS))
CLIOBJS8  := $(call obj,8,$(CLISRCS))
LIBOBJS10 := $(call obj,10,$(LIBSRCS))
LIBOBJS12 := $(call obj,12,$(LIBSRCS))
CLIOBJS10 := $(call obj,10,$(CLISRCS))
CLIOBJS12 := $(call obj,12,$(CLISRCS))

